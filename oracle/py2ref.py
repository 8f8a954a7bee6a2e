"""Run functions of the reference's Python-2 modules under Python 3 -- test infrastructure only.

The reference's callers (`get_train_data.py`, `roc_functions.py`, `edge_signal_proc.py`, ...) are Python 2: `print`
statements, `xrange`, classic `/` on ints.  They cannot be imported here (SyntaxError), and there is no Python 2
interpreter.  `load(path, names)` reads a module's source where it lies under /root/reference (nothing is copied
into the repo), and makes the named functions -- plus every module-level function they call -- runnable:

  1. `lib2to3` (ships with CPython) rewrites the Python-2-only syntax: print statements, xrange, dict methods, ...
  2. every `a / b` becomes `_py2div(a, b)`: floor division when both operands are integers (Python ints, NumPy
     integer scalars or integer arrays), true division otherwise -- Python 2's classic division;
  3. unary minus on a boolean array is logical not, as in the NumPy of the reference's time (`-mask`; NumPy >= 1.13
     raises) -- `_oldnp_neg`.  The same NumPy accepted integral floats where it wants integers today: `np.tile`
     repetitions are cast by the `np` the functions see (`_OldNumpy`); slice bounds cannot be intercepted, so
     callers of this module pass such arguments (orientation steps, row breaks) as ints.
  4. only the requested function definitions are executed, in a namespace that holds NumPy and whatever the
     caller passes in `env` (e.g. the reference's compiled Cython modules from oracle/_ref), so the module's own
     imports (matplotlib, cPickle, ...) are never needed.

Used by tests/golden/make_golden_*.py to produce golden vectors and by tests/test_oracle.py to check the CPU
restatements against the reference's own code on fresh inputs.  Returns None when /root/reference is absent
(the GPU box).
"""
import ast
import os

import numpy as np

REFERENCE = "/root/reference/template_speech_rec"


def _py2div(a, b):
    """Python 2's `/`: floor division for integer operands, true division otherwise."""
    def is_int(x):
        if isinstance(x, (bool, int, np.integer)):
            return True
        return isinstance(x, np.ndarray) and x.dtype.kind in "iub"
    if is_int(a) and is_int(b):
        return np.floor_divide(a, b) if isinstance(a, np.ndarray) or isinstance(b, np.ndarray) else a // b
    return a / b


def _oldnp_neg(x):
    """Unary minus of NumPy < 1.13: logical not on boolean arrays."""
    if isinstance(x, (np.ndarray, np.bool_)) and np.asarray(x).dtype == np.bool_:
        return np.logical_not(x)
    return -x


class _OldNumpy(object):
    """`np` as the reference's functions see it: NumPy, with integral-float repetition counts accepted by tile."""
    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def tile(A, reps):
        reps = tuple(int(r) for r in reps) if isinstance(reps, (tuple, list, np.ndarray)) else int(reps)
        return np.tile(A, reps)


class _ClassicDivision(ast.NodeTransformer):
    def visit_UnaryOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.USub) and not isinstance(node.operand, ast.Constant):
            return ast.copy_location(
                ast.Call(func=ast.Name(id="_oldnp_neg", ctx=ast.Load()), args=[node.operand], keywords=[]), node)
        return node

    def visit_BinOp(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):
            return ast.copy_location(
                ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()), args=[node.left, node.right], keywords=[]), node)
        return node

    def visit_AugAssign(self, node):
        self.generic_visit(node)
        if isinstance(node.op, ast.Div):  # a /= b  ->  a = _py2div(a, b)
            load = ast.parse(ast.unparse(node.target), mode="eval").body
            return ast.copy_location(
                ast.Assign(targets=[node.target],
                           value=ast.Call(func=ast.Name(id="_py2div", ctx=ast.Load()), args=[load, node.value],
                                          keywords=[])), node)
        return node


def _to_python3(source):
    from lib2to3 import refactor
    fixers = refactor.get_fixers_from_package("lib2to3.fixes")
    tool = refactor.RefactoringTool(fixers)
    if not source.endswith("\n"):
        source += "\n"
    return str(tool.refactor_string(source, "<reference>"))


def load(module, names, env=None):
    """-> {name: function or class} for `names` of REFERENCE/<module>.py (and the module-level functions and
    classes they use), or None when the reference is not there."""
    path = os.path.join(REFERENCE, module + ".py")
    if not os.path.exists(path):
        return None
    with open(path, encoding="latin-1") as f:
        tree = ast.parse(_to_python3(f.read()))
    defs = {n.name: n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef))}
    wanted, todo = [], list(names)
    while todo:
        n = todo.pop()
        if n in wanted:
            continue
        if n not in defs:
            raise KeyError("%s has no function %s" % (path, n))
        wanted.append(n)
        for node in ast.walk(defs[n]):
            if isinstance(node, ast.Name) and node.id in defs and node.id not in wanted:
                todo.append(node.id)
    body = [_ClassicDivision().visit(defs[n]) for n in defs if n in wanted]  # module order
    mod = ast.fix_missing_locations(ast.Module(body=body, type_ignores=[]))
    ns = {"np": _OldNumpy(), "numpy": _OldNumpy(), "_py2div": _py2div, "_oldnp_neg": _oldnp_neg}
    ns.update(env or {})
    exec(compile(mod, path, "exec"), ns)
    return {n: ns[n] for n in wanted}
