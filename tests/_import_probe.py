"""Helper run in a subprocess by tests/test_dropin_callers_cpu.py: pulls every `compression_method.*` / `token_compression.*`
import statement out of one of the reference's caller scripts (ast, no execution of the script itself) and executes
those statements with dropin/ ahead of the reference's package root.  Prints one JSON line."""
import ast
import importlib.util
import json
import os
import sys
import traceback

sys.dont_write_bytecode = True            # never drop __pycache__ into the reference tree
PKGS = ("compression_method", "token_compression")


def main():
    repo, caller, roots = sys.argv[1], sys.argv[2], sys.argv[3:]
    sys.path[:0] = [os.path.join(repo, "visionselector_amd", "dropin"), repo] + roots
    tree = ast.parse(open(caller).read())
    rows = []
    for node in ast.walk(tree):
        if not (isinstance(node, ast.ImportFrom) and node.level == 0 and node.module
                and node.module.split(".")[0] in PKGS):
            continue
        src = ast.unparse(node)
        row = {"stmt": src, "line": node.lineno, "module": node.module}
        try:
            row["origin"] = importlib.util.find_spec(node.module).origin
        except Exception as e:            # parent not importable / module missing
            row["origin"] = None
            row["find_error"] = f"{type(e).__name__}: {e}"
        ns = {}
        try:
            exec(src, ns)
            row["ok"] = True
            row["names"] = {k: (getattr(v, "__module__", None) or "") for k, v in ns.items()
                            if k != "__builtins__" and not k.startswith("__")}
        except BaseException as e:
            row["ok"] = False
            row["error"] = f"{type(e).__name__}: {e}"
            row["raised_in"] = traceback.extract_tb(e.__traceback__)[-1].filename
        rows.append(row)
    print("PROBE " + json.dumps(rows))


if __name__ == "__main__":
    main()
