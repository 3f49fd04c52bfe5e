"""The fixture generator must reach the REFERENCE, never this repository's own `optimal_crowds/` alias.

`tests/golden/make_goldens.py` imports `/root/reference` from a staging directory (reference simulations.py:10-11 imports
`from optimal_crowds import pedestrians / optimals`, so the staged directory carries that name).  The repository root holds
a regular package of the same name -- the drop-in alias of the MI355X implementation -- which Python prefers over the
staged namespace portion whenever the root is on `sys.path`: the generator would then record the implementation's own
output as "reference" fixtures and every parity test would compare the product with itself.  These tests run the
generator's staging in a child process started from the worst place (cwd = the repository root, root on PYTHONPATH, the
alias already imported) and check where the three modules came from.  Skipped where the reference is absent (GPU box).
"""
import os
import subprocess
import sys
import textwrap

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"
GEN = os.path.join(REPO, "tests", "golden", "make_goldens.py")

needs_reference = pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference only exists in the build container")


def _run(code, tmp_path):
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], cwd=REPO, env=env, capture_output=True,
                          text=True, timeout=300)


@needs_reference
def test_stage_imports_the_reference_not_the_alias(tmp_path):
    r = _run("""
        import importlib.util, os, sys
        sys.path.insert(0, %r)                 # the worst case: root first on the path ...
        try:
            import optimal_crowds              # ... and the alias already imported (fails without the built library: fine)
        except Exception:
            pass
        spec = importlib.util.spec_from_file_location("make_goldens", %r)
        mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
        mods = mg.stage(stage_dir=%r)
        for m in mods:
            f = os.path.realpath(m.__file__)
            assert f.startswith(%r + os.sep), f
            print("OK", m.__name__, f)
        import optimal_crowds.optimals as o2, optimal_crowds.pedestrians as p2
        assert o2 is mods[1] and p2 is mods[2]         # what reference simulations.py:10-11 binds is the reference too
        assert not any(os.path.realpath(p) == %r for p in sys.path if p)
    """ % (REPO, GEN, str(tmp_path / "stage"), os.path.realpath(REFERENCE), os.path.realpath(REPO)), tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("OK optimal_crowds.") == 3, r.stdout + r.stderr


@needs_reference
def test_generator_refuses_to_write_when_the_modules_are_not_the_reference(tmp_path):
    r = _run("""
        import importlib.util, os, sys, types
        spec = importlib.util.spec_from_file_location("make_goldens", %r)
        mg = importlib.util.module_from_spec(spec); spec.loader.exec_module(mg)
        fake = types.ModuleType("optimal_crowds.simulations"); fake.__file__ = os.path.join(%r, "optimal_crowds_amd", "simulations.py")
        mg.OUT = %r
        try:                                   # before stage(): nothing may be written
            mg.save("X.npz", a=1)
            print("WROTE")
        except RuntimeError as e:
            print("REFUSED-1", e)
        mg._MODS = (fake,)
        try:                                   # modules from the repository: nothing may be written
            mg.save("X.npz", a=1)
            print("WROTE")
        except RuntimeError as e:
            print("REFUSED-2", e)
        try:                                   # a staged "reference" that is not /root/reference is refused as well
            mg.check_is_reference((fake,), %r)
            print("ACCEPTED")
        except RuntimeError as e:
            print("REFUSED-3", e)
        assert not os.path.exists(os.path.join(mg.OUT, "X.npz"))
    """ % (GEN, REPO, str(tmp_path), REFERENCE), tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "WROTE" not in r.stdout and "ACCEPTED" not in r.stdout
    assert all(("REFUSED-%d" % i) in r.stdout for i in (1, 2, 3)), r.stdout


def test_generator_never_puts_the_repository_root_on_the_path():
    """static check, runs everywhere: the script must not import the repository as a package"""
    src = open(GEN).read()
    assert "sys.path.insert(0, REPO)" not in src and "from optimal_crowds_amd" not in src
    assert "check_is_reference((simulations, optimals, pedestrians), reference)" in src
