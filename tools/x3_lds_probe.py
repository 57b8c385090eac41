#!/usr/bin/env python3
"""Timing probe for k_conv_stack_x3 (results of the probe builds are WRONG by construction): how much of the kernel's time on real
operands is the LDS traffic of the activation fragments?  Builds libckr with -DCKR_X3_PROBE=1 (the fragments of the six taps with
dx != 0 are copies of the dx = 0 fragments: 2/3 of the ds_read_b128 gone, nothing in their place) and =2 (derived through a DPP wave
shift + mask per register: the price of making them in registers) and =3 (every other pair of weight fragments is a register copy: half
the L2 -> register weight stream) and =4 / 5 / 6 (other orders of the three MFMAs per multiply-add) and =7 (the operands in each other's MFMA slots), and times the float32-grade conv stack on 4 096 boards of random
planes with each.  The variants are kept as a patch (tools/x3_probes.patch) that this script applies to a temporary copy of the sources.

    python tools/x3_lds_probe.py build     # here (hipcc cross-compiles)
    python tools/x3_lds_probe.py run       # on the GPU: one JSON line per build"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "x3probe")
PROBED_REVISION = "a234e69"          # the commit that moved the probes out of csrc/ckr_conv_x3.hip: tools/x3_probes.patch applies to its file
sys.path.insert(0, ROOT)


def build():
    """The probe variants live in tools/x3_probes.patch, not in the product source: csrc/ is copied next to a copy of include/,
    the patch applied there, and one library built per probe."""
    import shutil, tempfile
    from checkers_mcts_amd import build as ckbuild
    os.makedirs(OUT, exist_ok=True)
    tmp = tempfile.mkdtemp(prefix="x3probe_")
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    shutil.copytree(ckbuild.CSRC, os.path.join(tmp, "checkers-mcts_amd", "csrc"))
    # the probes were written for the 208-VGPR kernel of rounds 2-4 (both activation-fragment halves double-buffered): the patch
    # applies to that revision of the conv stack's source, taken from the repository's history; the other sources are today's
    x3 = subprocess.check_output(["git", "-C", ROOT, "show", PROBED_REVISION + ":checkers-mcts_amd/csrc/ckr_conv_x3.hip"])
    open(os.path.join(tmp, "checkers-mcts_amd", "csrc", "ckr_conv_x3.hip"), "wb").write(x3)
    subprocess.check_call(["patch", "-p1", "-d", tmp, "-i", os.path.join(ROOT, "tools", "x3_probes.patch")])
    srcs = [os.path.join(tmp, "checkers-mcts_amd", "csrc", os.path.basename(f)) for f in ckbuild.sources()]
    for k in (0, 1, 2, 3, 4, 5, 6, 7):
        extra = ["-DCKR_X3_PROBE=%d" % k] if k else []
        subprocess.check_call([ckbuild.HIPCC] + ckbuild.FLAGS + extra + srcs + ["-o", os.path.join(OUT, "libckr_probe%d.so" % k)])
    shutil.rmtree(tmp)


def run_one(k):
    from checkers_mcts_amd import _lib
    _lib.LIB_PATH = os.path.join(OUT, "libckr_probe%d.so" % k)
    import torch
    from checkers_mcts_amd import net as N
    from checkers_mcts_amd.fused import FusedEvaluator
    S = 4096
    m = N.PolicyValueNet(128).keras_init(0).eval().cuda()
    fe = FusedEvaluator(m, S, mode="f16x3")
    x = (torch.rand(S, 8, 8, 14, device="cuda") < 0.2).float().contiguous()
    for _ in range(20):
        fe.conv_only(x)
    torch.cuda.synchronize()
    out = []
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            fe.conv_only(x)
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / 200 * 1e3)
    print(json.dumps(dict(probe=k, us_per_launch=[round(v, 1) for v in out])), flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "run":
        for k in (0, 1, 2, 3, 4, 5, 6, 7, 0):
            subprocess.call([sys.executable, os.path.abspath(__file__), "one", str(k)])
    else:
        run_one(int(sys.argv[2]))
