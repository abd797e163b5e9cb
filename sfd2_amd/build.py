"""Builds libsfd2hip.so (hipcc, gfx950) in-tree.  Called by __graft_entry__.build()."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsfd2hip.so")
SOURCES = ["conv_kernels.hip", "conv2_kernels.hip", "conv3_kernels.hip", "conv3rf_kernels.hip", "conv2b_s2d_kernel.hip", "conv1x1_kernels.hip", "resblock_kernel.hip", "conv_f32_kernels.hip", "convc_kernels.hip", "rb23_c_kernel.hip", "sparse_da3_kernel.hip", "fused_stem_kernel.hip", "fused_stem_c_kernel.hip", "post_kernels.hip", "util_kernels.hip", "nms4_kernels.hip", "match_kernels.hip", "match_mutual_kernel.hip", "api_core.hip", "api_weights.hip", "api_network.hip", "api_extract.hip", "api_match.hip", "api_graph.hip"]
# per-source extra flags (see the header comment of each file)
# -fno-honor-nans: without it hipcc puts a NaN-canonicalising v_max_f32 x, x in front of every fmaxf operand it cannot
# prove quiet (the ReLUs of the epilogues: two VALU per value instead of one).  Finite inputs give identical results.
_NN = ["-fno-honor-nans"]
SRC_FLAGS = {"match_mutual_kernel.hip": _NN, "conv3_kernels.hip": _NN, "conv3rf_kernels.hip": _NN, "resblock_kernel.hip": _NN, "conv2_kernels.hip": _NN,
             "fused_stem_kernel.hip": _NN, "conv_kernels.hip": _NN, "conv1x1_kernels.hip": _NN, "nms4_kernels.hip": _NN, "convc_kernels.hip": _NN, "fused_stem_c_kernel.hip": _NN, "rb23_c_kernel.hip": _NN, "sparse_da3_kernel.hip": _NN, "conv2b_s2d_kernel.hip": _NN}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("SFD2_EXTRA_FLAGS", "").split()


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _hipcc():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def have_hipcc():
    return os.path.isfile(_hipcc()) and os.access(_hipcc(), os.X_OK)


def needs_build():
    """True when libsfd2hip.so is missing or older than any of its sources."""
    if not os.path.exists(LIB):
        return True
    srcs = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "sfd2_internal.h"), os.path.join(CSRC, "sfd2_ctx.h"),
                                                      os.path.join(HERE, "..", "include", "sfd2_hip.h")]
    return any(os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(LIB) for s in srcs)


def build_lib(force=False, verbose=False, out=None, extra_flags=()):
    """out / extra_flags: experiment builds (e.g. -DSFD2_EXPERIMENTS into another .so); objects then go to a
    side directory so the product objects are not disturbed."""
    hipcc = _hipcc()
    objs = []
    deps = [os.path.join(CSRC, "sfd2_internal.h"), os.path.join(CSRC, "sfd2_ctx.h"), os.path.join(HERE, "..", "include", "sfd2_hip.h")]
    procs = []
    lib = out or LIB
    odir = CSRC
    if out:
        odir = os.path.join(os.path.dirname(os.path.abspath(out)), "obj_" + os.path.basename(out).replace(".", "_"))
        os.makedirs(odir, exist_ok=True)
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(odir, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [hipcc] + FLAGS + SRC_FLAGS.get(src, []) + list(extra_flags) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or procs or not os.path.exists(lib) or any(_newer(o, lib) for o in objs):
        tmp = lib + ".tmp.%d" % os.getpid()
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(tmp, lib)   # atomic: a process loading the library meanwhile sees the old or the new file, never a partial one
    return lib


if __name__ == "__main__":
    print(build_lib(verbose=True))
