"""Builds libilqr_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib", "libilqr_amd.so")
SOURCES = [os.path.join(CSRC, "capi.hip")]
import glob


def _headers():
    """Everything the library is compiled from besides SOURCES: every file under csrc/ (globbed, so a header added
    later is hashed without anyone remembering to list it) and the public C headers."""
    inc = os.path.join(os.path.dirname(PKG), "include")
    return sorted(set(glob.glob(os.path.join(CSRC, "*")) + glob.glob(os.path.join(inc, "*.h"))) - set(SOURCES))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: the SLP vectoriser turns the float dot products of the Riccati step into v_pk_mul_f32 +
# scalar adds instead of FMA chains (fp32 backward kernel 0.64 -> 0.60 ms); there is no packed fp64 arithmetic
# for it to find, so the fp64 kernels do not change.
# (-ffp-contract stays at hipcc's default: =on measured 2-10 % slower.  Candidate states are re-integrated from
# checkpoints by several kernels; tests/test_gpu_fused_sweep.py::test_committed_trajectory_is_the_rollout_that_was_scored
# pins that they all reproduce the scored rollout bit for bit.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-slp-vectorize"]


HASHFILE = LIB + ".srchash"


def _source_hash():
    import hashlib
    h = hashlib.sha256()
    for f in sorted(SOURCES + _headers()):
        if os.path.exists(f):
            h.update(os.path.basename(f).encode())
            h.update(open(f, "rb").read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def stale():
    """True when the library is missing or was built from different sources (content hash, not
    mtimes: the tree is copied to the GPU box and 8 ranks may import it at once)."""
    if not os.path.exists(LIB) or not os.path.exists(HASHFILE):
        return True
    return open(HASHFILE).read().strip() != _source_hash()


def build(force=False, verbose=False):
    """Compile the extension if missing or built from other sources. Returns the .so path."""
    if force or stale():
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        tmp = LIB + ".tmp.%d" % os.getpid()
        cmd = [HIPCC] + FLAGS + ["-o", tmp] + SOURCES
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, LIB)  # atomic: concurrent importers never see a half-written library
        with open(HASHFILE, "w") as f:
            f.write(_source_hash())
    return LIB


def build_user(header, out, verbose=False):
    """A build of the library that carries the caller's device twin (ILQR_MODEL_USER): `header` defines
    UserModelT<real> (contract: csrc/models.hpp).  Rebuilt when the library sources, the header or the flags change."""
    import hashlib
    header, out = os.path.abspath(header), os.path.abspath(out)
    h = hashlib.sha256((_source_hash() + open(header, "rb").read().hex()).encode()).hexdigest()
    stamp = out + ".srchash"
    if os.path.exists(out) and os.path.exists(stamp) and open(stamp).read().strip() == h:
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tmp = out + ".tmp.%d" % os.getpid()
    cmd = [HIPCC] + FLAGS + ["-DILQR_USER_MODEL_HEADER=\"%s\"" % header, "-o", tmp] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, out)
    with open(stamp, "w") as f:
        f.write(h)
    return out


USER_EXAMPLE_HEADER = os.path.join(os.path.dirname(PKG), "examples", "user_model_acrobot.hpp")
USER_EXAMPLE_LIB = os.path.join(PKG, "lib", "libilqr_amd_user_example.so")
# a user twin with dimensions of its own (n = 6, m = 2): the generic kernels
USER_EXAMPLE6_HEADER = os.path.join(os.path.dirname(PKG), "examples", "user_model_linear6.hpp")
USER_EXAMPLE6_LIB = os.path.join(PKG, "lib", "libilqr_amd_user_linear6.so")
# a user twin that is neither small nor linear-quadratic (n = 16, m = 4: a chain of coupled pendulums): the generic kernels, point-by-point finite differences
USER_CHAIN_HEADER = os.path.join(os.path.dirname(PKG), "examples", "user_model_pendulum_chain.hpp")
USER_CHAIN_LIB = os.path.join(PKG, "lib", "libilqr_amd_user_chain.so")
USER_BUILDS = ((USER_EXAMPLE_HEADER, USER_EXAMPLE_LIB), (USER_EXAMPLE6_HEADER, USER_EXAMPLE6_LIB), (USER_CHAIN_HEADER, USER_CHAIN_LIB))


def build_all(verbose=False):
    """The stock library and every example twin, compiled side by side (each is one translation unit of about two minutes)."""
    from concurrent.futures import ThreadPoolExecutor
    jobs = [lambda: build(force=True, verbose=verbose)] + [(lambda h=h, o=o: build_user(h, o, verbose=verbose)) for h, o in USER_BUILDS]
    with ThreadPoolExecutor(max_workers=len(jobs)) as ex:
        return [f.result() for f in [ex.submit(j) for j in jobs]]


if __name__ == "__main__":
    print(build(force=True, verbose=True))
