"""Oracle (test infrastructure only): the rotation-format helpers the reference's trainers call either side of the RVQ-VAEs, restated on
CPU torch from utils/rotation_conversions.py (a copy of pytorch3d's transforms), and the two places that use them:
`_load_data` (diffusion_rvqvae_trainer.py:244-295: axis-angle -> 6D per body part, normalise, `map2latent`, / latent scale) and the tail
of `_g_test` (:503-531: 6D -> axis-angle per part, scatter into the 165 axis-angle channels, jaw from the target, back to 6D).
Pinned to the reference's own `_load_data` / `_g_test` run by tests/golden/make_longform_golden.py (loaddata_outputs.npz, longform_outputs.npz)."""
import torch


def axis_angle_to_quaternion(aa):                          # rotation_conversions.py:448-477
    angles = torch.norm(aa, p=2, dim=-1, keepdim=True)
    half = 0.5 * angles
    small = angles.abs() < 1e-6
    s = torch.where(small, 0.5 - angles * angles / 48, torch.sin(half) / torch.where(small, torch.ones_like(angles), angles))
    return torch.cat([torch.cos(half), aa * s], dim=-1)


def quaternion_to_matrix(q):                               # :36-64
    r, i, j, k = torch.unbind(q, -1)
    two_s = 2.0 / (q * q).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(q.shape[:-1] + (3, 3))


def axis_angle_to_matrix(aa):                              # :416-430
    return quaternion_to_matrix(axis_angle_to_quaternion(aa))


def matrix_to_rotation_6d(m):                              # :535-550
    return m[..., :2, :].clone().reshape(*m.size()[:-2], 6)


def rotation_6d_to_matrix(d6):                             # :511-533
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = torch.nn.functional.normalize(b2, dim=-1)
    return torch.stack((b1, b2, torch.cross(b1, b2, dim=-1)), dim=-2)


def _sqrt_positive_part(x):                                # :85-93
    return torch.where(x > 0, torch.sqrt(x.clamp_min(0)), torch.zeros_like(x))


def _copysign(a, b):                                       # :67-82
    return torch.where((a < 0) != (b < 0), -a, a)


def matrix_to_quaternion(m):                               # :96-118
    m00, m11, m22 = m[..., 0, 0], m[..., 1, 1], m[..., 2, 2]
    o0 = 0.5 * _sqrt_positive_part(1 + m00 + m11 + m22)
    x = 0.5 * _sqrt_positive_part(1 + m00 - m11 - m22)
    y = 0.5 * _sqrt_positive_part(1 - m00 + m11 - m22)
    z = 0.5 * _sqrt_positive_part(1 - m00 - m11 + m22)
    return torch.stack((o0, _copysign(x, m[..., 2, 1] - m[..., 1, 2]), _copysign(y, m[..., 0, 2] - m[..., 2, 0]),
                        _copysign(z, m[..., 1, 0] - m[..., 0, 1])), -1)


def quaternion_to_axis_angle(q):                           # :480-508
    norms = torch.norm(q[..., 1:], p=2, dim=-1, keepdim=True)
    half = torch.atan2(norms, q[..., :1])
    angles = 2 * half
    small = angles.abs() < 1e-6
    s = torch.where(small, 0.5 - angles * angles / 48, torch.sin(half) / torch.where(small, torch.ones_like(angles), angles))
    return q[..., 1:] / s


def matrix_to_axis_angle(m):                               # :432-446
    return quaternion_to_axis_angle(matrix_to_quaternion(m))


def aa_to_6d(aa):
    return matrix_to_rotation_6d(axis_angle_to_matrix(aa))


def d6_to_aa(d6):
    return matrix_to_axis_angle(rotation_6d_to_matrix(d6))


def load_data_ref(pose165, trans_v, masks, stats, vq_sds, latent_scale=5.0):
    """diffusion_rvqvae_trainer.py:255-294: pose165 (B, n, 165) axis-angle, trans_v (B, n, 3); masks: 0/1 arrays over the 165 channels;
    stats: {"upper": (mean, std), ...,"trans": (mean, std)}; vq_sds: state dicts of the three RVQ-VAEs -> dict like the reference's."""
    from oracle import rvq_ref as rr
    bs, n, _ = pose165.shape
    part = lambda m, j: aa_to_6d(pose165[:, :, torch.as_tensor(m).bool()].reshape(bs, n, j, 3)).reshape(bs, n, j * 6)
    upper, lower = part(masks["upper"], 13), part(masks["lower"], 9)
    hands = aa_to_6d(pose165[:, :, 75:165].reshape(bs, n, 30, 3)).reshape(bs, n, 180)          # :262-264
    upper = (upper - stats["upper"][0]) / stats["upper"][1]                                    # :279-283
    hands = (hands - stats["hands"][0]) / stats["hands"][1]
    lower = (lower - stats["lower"][0]) / stats["lower"][1]
    lower = torch.cat([lower, (trans_v - stats["trans"][0]) / stats["trans"][1]], dim=-1)      # :285-287
    lat = [rr.map2latent(vq_sds[k], v) for k, v in (("upper", upper), ("hands", hands), ("lower", lower))]
    return {"tar_pose_upper": upper, "tar_pose_hands": hands, "tar_pose_lower": lower, "latent_in": torch.cat(lat, dim=2) / latent_scale,
            "tar_pose_6d": aa_to_6d(pose165.reshape(bs, n, 55, 3)).reshape(bs, n, 330)}        # :294, :297-298


def assemble_pose_ref(rec_upper, rec_hands, rec_lower, tar_pose165, masks):
    """diffusion_rvqvae_trainer.py:503-531: de-normalised 6D parts (B, n, 78 / 180 / 54) + the target's axis-angle pose -> rec_pose (B, n, 330)."""
    bs, n, _ = rec_upper.shape
    rec = torch.zeros(bs * n, 165)
    for x, j, m in ((rec_upper, 13, masks["upper"]), (rec_lower[..., :54], 9, masks["lower"]), (rec_hands, 30, masks["hands"])):
        idx = torch.where(torch.as_tensor(m) == 1)[0]                                          # inverse_selection_tensor (:236-242)
        rec[:, idx] = rec[:, idx] + d6_to_aa(x.reshape(bs, n, j, 6)).reshape(bs * n, j * 3)
    rec[:, 66:69] = tar_pose165.reshape(bs * n, 165)[:, 66:69]                                 # :526
    return aa_to_6d(rec.reshape(bs * n, 55, 3)).reshape(bs, n, 330)                            # :528-529
