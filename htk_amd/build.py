"""Build libhtk_amd.so (HIP kernels for gfx950 + host C) in-tree.

    python -m htk_amd.build          # or __graft_entry__.build()

hipcc cross-compiles gfx950 without a GPU.  Host C is compiled by gcc as C, kernels and launchers by
hipcc; everything is linked into ONE shared object next to this file so that it travels with the tree.
FMA contraction is off everywhere: the exact-order kernels must round like the reference's SSE2 code.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libhtk_amd.so")
HIP_SRCS = ["csrc/model.hip", "csrc/gmm_exact.hip", "csrc/gmm_mfma.hip", "csrc/gmm_bf16.hip", "csrc/gmm_f16.hip", "csrc/fb_kernels.hip", "csrc/fb_wave.hip", "csrc/fb_state.hip", "csrc/fb_lr.hip", "csrc/fb.hip", "csrc/viterbi.hip", "csrc/decode.hip", "csrc/decode_n.hip", "csrc/decode_ord.hip", "csrc/mfcc.hip", "csrc/update.hip", "csrc/comm.hip"]
C_SRCS = ["host/prep.c", "host/update.c", "host/fbank.c", "host/accio.c", "host/parmfile.c", "host/mmf.c", "host/labio.c", "host/net.c", "host/lattice.c"]
HEADERS = ["csrc/fb_lr_lean.inc", "csrc/internal.h", "csrc/kernels.h", "csrc/hipcheck.h", "csrc/ladd.h", "csrc/wavegrp.h", "csrc/fb_state.h", "csrc/decode.h", "csrc/decode_ord.h", "../include/htk_amd.h"]
ARCH = "gfx950"
# the tolerance-class scoring kernel never sees NaNs: lets v_max_f32 go without the IEEE canonicalisation of its operands
EXTRA_FLAGS = {"csrc/gmm_mfma.hip": ["-fno-honor-nans"], "csrc/gmm_bf16.hip": ["-fno-honor-nans"], "csrc/gmm_f16.hip": ["-fno-honor-nans"],
               "csrc/fb_kernels.hip": ["-Wno-pass-failed"]}      # (k_mixstate asks for an occupancy it knows it cannot have: MS_EU in fb_kernels.hip)
if os.environ.get("HTKAMD_LR_DEFS"):               # experiment switches of fb_lr.hip, e.g. HTKAMD_LR_DEFS="-DSTATS_EXP_NOOCC"
    EXTRA_FLAGS["csrc/fb_lr.hip"] = os.environ["HTKAMD_LR_DEFS"].split()
    EXTRA_FLAGS["csrc/fb_kernels.hip"] = EXTRA_FLAGS["csrc/fb_kernels.hip"] + os.environ["HTKAMD_LR_DEFS"].split()
if os.environ.get("HTKAMD_DEC_DEFS"):              # ... and decode.hip (-DDEC_CLK: phase stamps)
    EXTRA_FLAGS["csrc/decode.hip"] = os.environ["HTKAMD_DEC_DEFS"].split()
if os.environ.get("HTKAMD_EX_DEFS"):               # ... and gmm_exact.hip
    EXTRA_FLAGS["csrc/gmm_exact.hip"] = os.environ["HTKAMD_EX_DEFS"].split()
if os.environ.get("HTKAMD_MFCC_DEFS"):             # ... and mfcc.hip
    EXTRA_FLAGS["csrc/mfcc.hip"] = os.environ["HTKAMD_MFCC_DEFS"].split()
if os.environ.get("HTKAMD_UPD_DEFS"):              # the same for update.hip / gmm_bf16.hip (tools/r05_updvar.sh)
    EXTRA_FLAGS["csrc/update.hip"] = os.environ["HTKAMD_UPD_DEFS"].split()
if os.environ.get("HTKAMD_B16_CT"):              # experiment switch: column tiles per wavefront of the bf16 scoring kernel (gmm_bf16.hip: B16_COL_TILES)
    EXTRA_FLAGS["csrc/gmm_bf16.hip"] = EXTRA_FLAGS["csrc/gmm_bf16.hip"] + ["-DB16_COL_TILES=" + os.environ["HTKAMD_B16_CT"]]
if os.environ.get("HTKAMD_B16_TF"):              # experiment switch: frames per task of the bf16 scoring kernel in forward-backward (kernels.h: B16_TASK_FRAMES)
    EXTRA_FLAGS["csrc/gmm_bf16.hip"] = EXTRA_FLAGS["csrc/gmm_bf16.hip"] + ["-DB16_TASK_FRAMES=" + os.environ["HTKAMD_B16_TF"]]
    EXTRA_FLAGS["csrc/fb.hip"] = ["-DB16_TASK_FRAMES=" + os.environ["HTKAMD_B16_TF"]]
if os.environ.get("HTKAMD_B16_DEFS"):            # experiment switches of gmm_bf16.hip, e.g. HTKAMD_B16_DEFS="-DB16W_EU5=3"
    EXTRA_FLAGS["csrc/gmm_bf16.hip"] = EXTRA_FLAGS["csrc/gmm_bf16.hip"] + os.environ["HTKAMD_B16_DEFS"].split()
if os.environ.get("HTKAMD_B16_WPB"):             # experiment switch: wavefronts per workgroup of the bf16 scoring kernel (gmm_bf16.hip: B16_WPB)
    EXTRA_FLAGS["csrc/gmm_bf16.hip"] = EXTRA_FLAGS["csrc/gmm_bf16.hip"] + ["-DB16_WPB=" + os.environ["HTKAMD_B16_WPB"]]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS]
    objs = []
    procs = []
    for src in HIP_SRCS + C_SRCS:
        p = os.path.join(HERE, src)
        if not os.path.exists(p):
            continue
        o = os.path.join(OBJ, os.path.basename(src) + ".o")
        objs.append(o)
        if force or _newer(o, [p] + hdrs):
            if src.endswith(".hip"):
                cmd = ["hipcc", "--offload-arch=" + ARCH, "-O3", "-ffp-contract=off", "-fPIC", "-std=c++17",
                       "-Wall", "-Wno-unused-function"] + EXTRA_FLAGS.get(src, []) + ["-c", p, "-o", o]
            else:
                cmd = ["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fPIC", "-Wall", "-c", p, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise RuntimeError("build failed: " + " ".join(cmd))
    if force or procs or _newer(LIB, objs):
        cmd = ["hipcc", "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs + ["-lm", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_tools(verbose)
    return LIB


TOOLS = ["herest", "hvite"]


def build_tools(verbose: bool = False) -> None:
    """The command-line drivers (tools/*.c): plain C hosts over include/htk_amd.h, linked against the library in-tree."""
    root = os.path.dirname(HERE)
    bindir = os.path.join(root, "tools", "bin")
    os.makedirs(bindir, exist_ok=True)
    for t in TOOLS:
        src = os.path.join(root, "tools", t + ".c")
        if not os.path.exists(src):
            continue
        exe = os.path.join(bindir, t)
        if not _newer(exe, [src, os.path.join(root, "tools", "cli_common.h"), os.path.join(root, "include", "htk_amd.h"), LIB]):
            continue
        cmd = ["gcc", "-O2", "-std=gnu11", "-Wall", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "tools"), src, "-o", exe,
               "-L" + HERE, "-lhtk_amd", "-Wl,-rpath,$ORIGIN/../../htk_amd", "-lm"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
