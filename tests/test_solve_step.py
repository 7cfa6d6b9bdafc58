"""The device-resident dogleg step (csrc/solve_step.h) replayed on the CPU by its one-thread executor against the host solver
it replaces (host_solver.h: solve_dogleg): same iteration counts, accepted / rejected steps, termination codes and cost
traces on synthetic windows with and without a marginalization prior, free / fixed extrinsic, Wo in {2, 5, 7}.  Host code
only (no GPU); the device-only panel routines are checked by tests/test_gpu_dev_solver.py."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_solve_step_emulation_matches_host_solver(tmp_path):
    exe = str(tmp_path / "solve_step_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-std=c++17", "-ffp-contract=off", "-mavx2", "-Wno-unused-function",
                    "-I", os.path.join(ROOT, "lio-mapping_amd", "csrc"), os.path.join(ROOT, "tests", "host", "solve_step_check.hip"), "-o", exe],
                   check=True)
    for split in ("0", "1"):   # the host solver with and without the speed-bias block factored ahead (SplitFactor, opt-in)
        r = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, LIO_SPLIT_FACTOR=split))
        print(r.stdout)
        assert r.returncode == 0, r.stdout + r.stderr
        assert r.stdout.strip().endswith("OK")
        assert r.stdout.count("need_host=0") >= 15
