"""Build recipe for libckr.so (hand-written HIP for gfx950, C-ABI in include/ckr.h).

    python -m checkers_mcts_amd.build          # or __graft_entry__.build()

hipcc cross-compiles without a GPU; the .so is built IN-TREE
(checkers-mcts_amd/libckr.so) so it travels with the repo snapshot.
-ffp-contract=off: the PUCT / prior arithmetic must round exactly like the
reference's NumPy expressions (no fused multiply-add).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["ckr_rules.hip", "ckr_engine.hip", "ckr_conv.hip", "ckr_conv_x3.hip", "ckr_train.hip"]
LIB = os.path.join(HERE, "libckr.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: hipcc (ROCm 7.2) SLP-packs adjacent scalar float32 multiply-adds into v_pk_fma_f32 / v_pk_mul_f32
# with op_sel swizzles.  In the 1x1 policy head of ckr_conv_x3.hip that code returned wrong sums in lanes 48-63 of one
# accumulator for ~13 % of the boards -- only when two workgroups shared a CU (4 096-board batches; never at <= 1 023
# boards), bit-reproducible offsets, gone with this flag (tests/test_fullsize_gpu.py::
# test_fused_evaluator_full_batch_rows_vs_float64 is the regression test).  Packed f32 VALU is no gain beside MFMAs
# anyway (MI355X_MICROARCH.md, "price of one filler").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-fno-fast-math", "-fno-slp-vectorize", "-Wall", "-Wno-unused-function"]


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "ckr.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not is_stale():
        return LIB
    cmd = [HIPCC] + FLAGS + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
