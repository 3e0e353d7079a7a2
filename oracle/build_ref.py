"""Recipe for oracle/_ref: the REFERENCE's own matcher modules as byte-compiled, sourceless Python.

TEST / BENCH INFRASTRUCTURE ONLY.  The reference is Python, so "building" it means compiling the module
closure of ``gluefactory.models.matchers.lightglue`` from the sources WHERE THEY LIE under /root/reference into ``oracle/_ref/**.pyc`` -- outputs only; no
reference source is copied into the repository, and ``oracle/_ref/`` is git-ignored (it travels to the GPU box with
the snapshot, like the built libgf_amd.so).  Consumers: ``bench.py``'s ``cpu_baseline`` leg, which times the
reference's LightGlue train step on the GPU box's host cores (``"kind": "reference"``), and tests/test_gpu_reference_boundary.py,
where the reference's own TwoViewPipeline / TripletPipeline drive the HIP matchers on the GPU box.

    python oracle/build_ref.py        # needs /root/reference (build container); no-op message otherwise
"""
import importlib.util
import os
import py_compile
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "_ref")
STUBS = os.path.join(HERE, "stubs")          # omegaconf / kornia stand-ins (ours, tracked)
# Only what bench.py's cpu_baseline times: the LightGlue matcher and its losses.  (gluefactory_nonfree -- SuperGlue, under
# its own non-commercial licence -- is NOT bundled; SuperGlue / GlueStick goldens come from oracle/gen_golden.py, which
# imports the reference in place.)
TARGETS = ["gluefactory.models.matchers.lightglue", "gluefactory.models.utils.losses",
           "gluefactory.models.utils.metrics", "gluefactory.models.base_model", "gluefactory.models",
           # the CALLERS of the plugin boundary (two_view_pipeline.py:70-113, triplet_pipeline.py:23-99): the GPU tests drive
           # the HIP matchers through the reference's own pipeline code (tests/test_gpu_reference_boundary.py)
           "gluefactory.models.two_view_pipeline", "gluefactory.models.triplet_pipeline",
           # the train loop itself (train.py:216-683: dataset -> loader -> autocast -> GradScaler -> clip -> optimiser ->
           # scheduler -> validation -> checkpoint) driving the HIP matcher: tests/test_gpu_reference_train.py
           "gluefactory.train"]


def _closure(targets):
    """Files under /root/reference that the target modules import, found by reading their import statements (ast): the
    reference is NOT imported or executed by the build."""
    import ast

    def mod_file(mod):
        base = os.path.join(REF, *mod.split("."))
        for cand in (base + ".py", os.path.join(base, "__init__.py")):
            if os.path.isfile(cand):
                return cand
        return None

    def packages(mod):                     # importing a.b.c executes a/__init__ and a/b/__init__ first
        parts = mod.split(".")
        return [".".join(parts[:i]) for i in range(1, len(parts))]

    seen, todo = {}, list(targets)
    while todo:
        mod = todo.pop()
        f = mod_file(mod)
        if f is None or f in seen.values():
            continue
        seen[mod] = f
        todo += packages(mod)
        pkg = mod if f.endswith("__init__.py") else mod.rpartition(".")[0]
        with open(f) as fh:
            tree = ast.parse(fh.read(), f)
        for node in ast.walk(tree):
            if isinstance(node, ast.Import):
                todo += [a.name for a in node.names if a.name.split(".")[0] == "gluefactory"]
            elif isinstance(node, ast.ImportFrom):
                if node.level:
                    up = pkg.split(".")[:len(pkg.split(".")) - (node.level - 1)]
                    base = ".".join(up + ([node.module] if node.module else []))
                else:
                    base = node.module or ""
                if base.split(".")[0] != "gluefactory":
                    continue
                todo.append(base)
                todo += [base + "." + a.name for a in node.names]      # `from pkg import submodule`
    return sorted(set(seen.values()))


def build(verbose=True):
    if not os.path.isdir(os.path.join(REF, "gluefactory")):
        if verbose:
            print("oracle/build_ref.py: /root/reference not present; keeping the prebuilt oracle/_ref (if any)")
        return False
    files = _closure(TARGETS)
    shutil.rmtree(OUT, ignore_errors=True)
    for src in files:
        rel = os.path.relpath(src, REF)
        dst = os.path.join(OUT, rel[:-3] + ".pyc")      # sourceless layout: <pkg>/<module>.pyc
        os.makedirs(os.path.dirname(dst), exist_ok=True)
        py_compile.compile(src, cfile=dst, dfile=rel, doraise=True, optimize=0)
    with open(os.path.join(OUT, "MANIFEST.txt"), "w") as f:
        f.write("byte-compiled from /root/reference by oracle/build_ref.py with python %s magic %s\n"
                % (sys.version.split()[0], importlib.util.MAGIC_NUMBER.hex()))
        f.write("\n".join(os.path.relpath(s, REF) for s in files) + "\n")
    if verbose:
        print(f"oracle/_ref: {len(files)} reference modules byte-compiled")
    return True


def import_reference():
    """Put oracle/_ref (+ the stand-ins) on sys.path; returns False when it has not been built or was byte-compiled by
    another Python (sourceless .pyc files only load under the interpreter version that wrote them)."""
    pyc = os.path.join(OUT, "gluefactory", "models", "matchers", "lightglue.pyc")
    if not os.path.exists(pyc):
        return False
    with open(pyc, "rb") as f:
        if f.read(4) != importlib.util.MAGIC_NUMBER:
            return False
    for p in (OUT, STUBS):
        if p not in sys.path:
            sys.path.insert(0, p)
    return True


if __name__ == "__main__":
    build()
