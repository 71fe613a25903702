"""Undefined-name check for the package's modules (no pyflakes in the image): every Name that is read must be bound in its function, an enclosing
function, the module, or builtins.  Usage: python scripts/lint_names.py file.py ..."""
import ast, builtins, sys


def bound_names(node):
    names = set()
    for n in ast.walk(node):
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.ClassDef)):
            names.add(n.name)
        if isinstance(n, (ast.FunctionDef, ast.AsyncFunctionDef, ast.Lambda)):
            a = n.args
            for x in a.posonlyargs + a.args + a.kwonlyargs:
                names.add(x.arg)
            if a.vararg: names.add(a.vararg.arg)
            if a.kwarg: names.add(a.kwarg.arg)
        elif isinstance(n, ast.Name) and isinstance(n.ctx, (ast.Store, ast.Del)):
            names.add(n.id)
        elif isinstance(n, (ast.Import, ast.ImportFrom)):
            for al in n.names:
                names.add((al.asname or al.name).split(".")[0])
        elif isinstance(n, ast.ExceptHandler) and n.name:
            names.add(n.name)
        elif isinstance(n, (ast.Global, ast.Nonlocal)):
            names.update(n.names)
    return names


def check(path):
    tree = ast.parse(open(path).read())
    mod = bound_names(tree) | set(dir(builtins)) | {"__file__", "__name__"}
    bad = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load) and n.id not in mod:
            bad.append((n.lineno, n.id))
    return bad


if __name__ == "__main__":
    rc = 0
    for p in sys.argv[1:]:
        for ln, name in check(p):
            print(f"{p}:{ln}: undefined name {name}")
            rc = 1
    sys.exit(rc)
