"""The C++ host facade (include/ilqr_amd.hpp): source compatibility with the reference's Model /
iLQR interface, loud failure without a GPU, and -- on the GPU box -- the two canonical problems
of the reference's driver (src/run_ilqr.cpp) end to end, including the CSV side effect."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "examples", "run_ilqr")
EIGEN = "/root/reference/include/eigen"


def build_example(out=EXE, extra=()):
    from ilqr_amd import _build
    _build.build()
    cmd = ["g++", "-std=c++14", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), *extra,
           os.path.join(ROOT, "examples", "run_ilqr.cpp"), "-o", out,
           "-L" + os.path.join(ROOT, "ilqr_amd", "lib"), "-lilqr_amd", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "ilqr_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return out


def test_facade_builds_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build_example()
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    r = subprocess.run([exe, "acrobot", "--quiet"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 2
    assert "no HIP device" in r.stderr


@pytest.mark.skipif(not os.path.isdir(EIGEN), reason="vendored Eigen only exists in the build container")
def test_facade_compiles_against_eigen_types(tmp_path):
    """With <Eigen/Core> on the include path the facade's Model/iLQR use Eigen::VectorXd, i.e. a
    user's Model subclass written for the reference compiles unchanged."""
    src = tmp_path / "user_model.cpp"
    src.write_text('''
#include "ilqr_amd.hpp"
#include <type_traits>
static_assert(std::is_same<ilqr_amd::VectorXd, Eigen::VectorXd>::value, "Eigen types expected");
// a Model written against include/model.h of the reference
class Pendulum : public ilqr_amd::Model {
 public:
  Pendulum() { x_dims = 2; u_dims = 1; u_min = Eigen::VectorXd::Constant(1, -2.0); u_max = Eigen::VectorXd::Constant(1, 2.0); }
  virtual Eigen::VectorXd dynamics(const Eigen::VectorXd& x, const Eigen::VectorXd& u) override {
    Eigen::VectorXd dx(2); dx << x(1), u(0) - 9.81 * sin(x(0)); return dx; }
  virtual double cost(const Eigen::VectorXd& x, const Eigen::VectorXd& u) override { return x.dot(x) + u.dot(u); }
  virtual double final_cost(const Eigen::VectorXd& x) override { return 10 * x.dot(x); }
};
int main() {
  Pendulum p; Eigen::VectorXd x(2); x << 0.1, 0.0; Eigen::VectorXd u(1); u << 0.5;
  Eigen::VectorXd x1 = p.integrate_dynamics(x, u, 0.02);
  if (std::abs(x1(0) - 0.1) > 1e-12) return 1;
  // the shipped LQ model: host evaluation of xdot = Ax + Bu, 0.5(x'Qx + u'Ru)
  ilqr_amd::LinearQuadratic lq(2, 1, {0, 1, -1, 0}, {0, 1}, {2, 0, 0, 2}, {4}, {2, 0, 0, 2}, -1, 1);
  Eigen::VectorXd dx = lq.dynamics(x, u);
  if (std::abs(dx(0) - 0.0) > 1e-15 || std::abs(dx(1) - (-0.1 + 0.5)) > 1e-15) return 4;
  if (std::abs(lq.cost(x, u) - 0.5 * (2 * 0.01 + 4 * 0.25)) > 1e-15 || lq.device_model_id() != ILQR_MODEL_LQ) return 5;
  try { ilqr_amd::iLQR s(new Pendulum(), 0.02); ilqr_amd::VecOfVecXd u0(5, u); s.verbose = false; s.init_traj(x, u0); }
  catch (const std::runtime_error& e) { return std::string(e.what()).find("no HIP device") != std::string::npos ? 0 : 2; }
  return 3;
}
''')
    exe = tmp_path / "user_model"
    subprocess.check_call(["g++", "-std=c++14", "-O1", "-w", "-I" + os.path.join(ROOT, "include"), "-I" + EIGEN, str(src),
                           "-o", str(exe), "-L" + os.path.join(ROOT, "ilqr_amd", "lib"), "-lilqr_amd", "-L/opt/rocm/lib",
                           "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "ilqr_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    # a host-only Model is accepted (its virtuals are evaluated on the host, the backward pass on the
    # device), but there is no path without a GPU: here ilqr_create fails loudly
    assert subprocess.run([str(exe)]).returncode == 0


@pytest.mark.gpu
def test_run_ilqr_acrobot(tmp_path):
    """`./run_iLQR acrobot`: 100 iterations, final cost 5.39788253688."""
    exe = build_example()
    r = subprocess.run([exe, "acrobot"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert "Initial cost: 3947.61" in out
    assert "iteration\tcost\t\treduction\texpect\t\tgrad\t\tlog10(lambda)" in out
    last = [l for l in out.splitlines() if l.startswith("final cost")][0].split()
    assert abs(float(last[2]) - 5.39788253688) < 1e-6 * 5.4 and int(last[4]) == 100 and int(last[6]) == 4
    # CSV side effect of every solve (ilqr_core.cpp:300, format :414-431)
    lines = (tmp_path / "ilqr_result.csv").read_text().split("\n")
    assert lines[0] == "x1, x2, x3, x4, u0, u1"
    assert len(lines) == 1 + 499 + 1 and lines[-1].endswith(", ") and lines[1].count(",") == 4
    # second printed row of the reference's progress table: cost 2.66e3 at iteration 1
    row1 = [l for l in out.splitlines() if l.startswith("1 ")][0].split()
    assert row1[1] == "2.66e+03"


@pytest.mark.gpu
def test_run_ilqr_integrator(tmp_path):
    exe = build_example()
    r = subprocess.run([exe, "integrator", "--quiet"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stderr
    last = [l for l in r.stdout.splitlines() if l.startswith("final cost")][0].split()
    assert abs(float(last[2]) - 356.168506469842) < 1e-6 * 356
    assert 5 <= int(last[4]) <= 15


@pytest.mark.gpu
@pytest.mark.parametrize("iters", [6, 0])
def test_host_only_model_matches_its_device_twin(tmp_path, iters):
    """examples/host_model.cpp: a user's Model subclass with no device twin (rollouts and finite
    differences through its host virtuals, backward pass / box-QP / accept logic on the GPU) takes
    the same iterations as the shipped device twin of the same model."""
    from ilqr_amd import _build
    _build.build()
    exe = str(tmp_path / "host_model")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-DILQR_AMD_NO_EIGEN", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "host_model.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "ilqr_amd", "lib"), "-lilqr_amd", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "ilqr_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe] + ([str(iters)] if iters else []), capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stderr
    rows = {l.split()[0]: l.split() for l in r.stdout.splitlines()}
    h, d = rows["host_model"], rows["device_twin"]
    c0h, ch, ith, sth = float(h[2]), float(h[4]), int(h[6]), int(h[8])
    c0d, cd, itd, std_ = float(d[2]), float(d[4]), int(d[6]), int(d[8])
    assert abs(c0h - 494.1509440000001) < 1e-9 and abs(c0d - c0h) < 1e-9      # SURVEY.md 8c anchor
    if iters:
        assert ith == itd == iters and sth == std_ == 0
        assert abs(ch - cd) < 1e-6 * abs(cd)
        assert float(rows["max_abs_diff"][2]) < 1e-6 and float(rows["max_abs_diff"][4]) < 1e-6
    else:  # solved to termination: the absolute stopping tests may tie one iteration apart
        assert sth > 0 and std_ > 0 and abs(ith - itd) <= 1
        assert abs(ch - cd) < 1e-4 * abs(cd) and abs(cd - 356.168506469842) < 1e-4 * 356


@pytest.mark.gpu
def test_host_model_evaluated_by_several_threads(tmp_path):
    """BatchILQR::set_host_threads: a batch of host-evaluated problems with the model's virtuals called from four
    OpenMP threads gives exactly what one thread gives (trajectories are independent; include/ilqr_amd.hpp)."""
    from ilqr_amd import _build
    _build.build()
    exe = str(tmp_path / "host_model_omp")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-fopenmp", "-DILQR_AMD_NO_EIGEN", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "host_model.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "ilqr_amd", "lib"), "-lilqr_amd", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "ilqr_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, "3", "4"], capture_output=True, text=True, cwd=tmp_path, timeout=300)
    assert r.returncode == 0, r.stderr
    row = [l.split() for l in r.stdout.splitlines() if l.startswith("host_threads")][0]
    assert int(row[1]) == 4 and float(row[3]) == 0.0


@pytest.mark.gpu
def test_user_device_twin_through_the_facade(tmp_path):
    """examples/user_model.cpp: a Model subclass whose device twin lives in a header OUTSIDE the library
    (examples/user_model_acrobot.hpp, compiled in by ilqr_amd._build.build_user) solved through ilqr_amd::iLQR, linked
    against that build: identical, to the last bit, to the shipped acrobot on the same problem."""
    from ilqr_amd import _build
    lib = _build.build_user(_build.USER_EXAMPLE_HEADER, _build.USER_EXAMPLE_LIB)
    exe = str(tmp_path / "user_model")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "user_model.cpp"), "-o", exe, lib, "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bit_identical 1" in r.stdout


@pytest.mark.gpu
def test_sharded_batch_from_cpp_equals_python_shards_and_one_handle(tmp_path):
    """examples/multi_gpu.cpp: ilqr_amd::ShardedBatchILQR, one C++ process, 8 logical shards (all on device 0 on a one-GPU
    box: the partition SURVEY.md 8e describes, the gather of the costs through ilqr_group_gather_costs) against the Python
    side's 8 separate handles and against ONE handle of the whole batch: costs and controls bit for bit, in global order;
    a ragged last shard too.  And the RCCL gather itself (ncclCommInitAll + ncclAllGather, loaded on demand), forced on one
    rank, returns the same vector."""
    import ctypes as C
    import numpy as np
    from ilqr_amd import BatchILQR, _build, capi
    from tests.util import acrobot_x0
    _build.build()
    exe = str(tmp_path / "multi_gpu")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "multi_gpu.cpp"), "-o", exe,
                           "-L" + os.path.join(ROOT, "ilqr_amd", "lib"), "-lilqr_amd", "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.join(ROOT, "ilqr_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    T, iters = 120, 3
    for B, ns in ((1024, 8), (203, 8)):
        x0 = acrobot_x0(B, scale=0.5, seed=21)
        x0.tofile(tmp_path / "x0.bin")
        r = subprocess.run([exe, str(tmp_path / "x0.bin"), str(B), str(T), str(ns), str(iters), str(tmp_path / "out.bin")],
                           capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "gather over copies" in r.stdout  # (one device: the shards are copied, not all-gathered)
        out = np.fromfile(tmp_path / "out.bin")
        cost_cpp, us_cpp = out[:B], out[B:].reshape(B, T, 1)
        # Python: the same partition as separate handles, and one handle of the whole batch
        per = (B + ns - 1) // ns
        cost_py, us_py = [], []
        for i in range(ns):
            lo, hi = i * per, min(B, (i + 1) * per)
            if lo >= hi:
                break
            g = BatchILQR("acrobot", hi - lo, T, 0.02, u_min=-1.5, u_max=1.5)
            g.init_traj(x0[lo:hi], np.zeros((hi - lo, T, 1)))
            g.iterate(iters)
            cost_py.append(g.cost())
            us_py.append(g.trajectory()[1])
            g.close()
        g = BatchILQR("acrobot", B, T, 0.02, u_min=-1.5, u_max=1.5)
        g.init_traj(x0, np.zeros((B, T, 1)))
        g.iterate(iters)
        assert np.array_equal(cost_cpp, np.concatenate(cost_py)) and np.array_equal(us_cpp, np.concatenate(us_py))
        assert np.array_equal(cost_cpp, g.cost()) and np.array_equal(us_cpp, g.trajectory()[1])
        # the RCCL route of the gather on this one device: a group of one shard with RCCL forced
        lib = g.lib
        hs = (C.c_void_p * 1)(g.h)
        grp = C.c_void_p()
        capi.check(lib.ilqr_group_create(hs, 1, 1, C.byref(grp)), lib)
        nr = C.c_int(0)
        assert lib.ilqr_group_uses_rccl(grp, C.byref(nr)) == 1 and nr.value == 1
        got = np.zeros(B)
        capi.check(lib.ilqr_group_gather_costs(grp, got.ctypes.data_as(C.POINTER(C.c_double))), lib)
        lib.ilqr_group_destroy(grp)
        assert np.array_equal(got, cost_cpp)
        g.close()


@pytest.mark.gpu
def test_user_twin_with_its_own_dimensions_against_host_virtuals(tmp_path):
    """examples/user_model_generic.cpp: a six-state Model solved through its device twin in the generic kernels
    (examples/user_model_linear6.hpp, ILQR_MODEL_USER), through its host virtuals (ILQR_MODEL_HOST) and through the shipped LQ
    twin: costs after three iterations within 1e-6 of each other."""
    from ilqr_amd import _build
    lib = _build.build_user(_build.USER_EXAMPLE6_HEADER, _build.USER_EXAMPLE6_LIB)
    exe = str(tmp_path / "user_model_generic")
    subprocess.check_call(["g++", "-std=c++14", "-O2", "-Wall", "-pthread", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "user_model_generic.cpp"), "-o", exe, lib, "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "agree 1" in r.stdout
    # ... and under ILQR_FLAG_REFERENCE_FIXES: the host-evaluated route's rollouts (the facade's) clamp as the device twins' do
    assert "fixes agree 1" in r.stdout, r.stdout
