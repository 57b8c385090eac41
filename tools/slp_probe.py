"""Root-causing the wrong sums of the SLP-vectorised 1x1 policy head of k_conv_stack_x3 (build.py: -fno-slp-vectorize).

    python tools/slp_probe.py build      # here (hipcc cross-compiles): variants of the kernel as gfx950 code objects in build/slp/
    python tools/slp_probe.py run        # on the GPU: every variant on the 4 096-board batch; JSON lines on stdout

`build` compiles csrc/ckr_conv_x3.hip to device assembly WITH the SLP vectoriser (the failing configuration) and without it,
then derives variants of the failing assembly by inserting instructions around the packed-float32 operations (s_nop = the
software wait states a missed hazard would need; s_waitcnt = what a missed LDS dependency would need), in the whole kernel
or in one cluster of packed operations at a time, and assembles each into a code object.  `run` launches each through
libckr's CKR_X3_CODE_OBJECT hook (the host code and every other kernel stay the same) and compares all 4 096 rows with a
float64 evaluation of the same network: which variant makes the error disappear names the mechanism."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "build", "slp")
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "checkers-mcts_amd", "csrc", "ckr_conv_x3.hip")
LLVM = "/opt/rocm/lib/llvm/bin"
BASE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-S", "--cuda-device-only"]
PK = re.compile(r"^\s*v_pk_(fma|mul|add)_f32\b")
KERNEL_END = re.compile(r"^\s*\.globl\s+_ZN4ckrp")             # the next kernel of the file: edits stop here


def compile_asm(name, extra):
    path = os.path.join(OUT, name + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + BASE_FLAGS + extra + ["-o", path, SRC], stderr=subprocess.DEVNULL)
    return path


def assemble(asm_path):
    obj, co = asm_path[:-2] + ".o", asm_path[:-2] + ".co"
    subprocess.check_call([LLVM + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", asm_path, "-o", obj])
    subprocess.check_call([LLVM + "/ld.lld", "-shared", obj, "-o", co])
    os.remove(obj)
    return co


def clusters(lines, gap=200):
    """Line ranges [a, b] of the first kernel in which packed-float32 operations cluster."""
    idx = []
    for i, ln in enumerate(lines):
        if KERNEL_END.match(ln):
            break
        if PK.match(ln):
            idx.append(i)
    out, start, prev = [], idx[0], idx[0]
    for i in idx[1:]:
        if i - prev > gap:
            out.append((start, prev)); start = i
        prev = i
    out.append((start, prev))
    return out


def edit(lines, where, before=None, after=None, match=PK):
    lo, hi = where
    out = []
    for i, ln in enumerate(lines):
        hit = lo <= i <= hi and match.match(ln)
        if hit and before:
            out.append("\t" + before + "\n")
        out.append(ln)
        if hit and after:
            out.append("\t" + after + "\n")
    return out


def build():
    os.makedirs(OUT, exist_ok=True)
    # the CKR_X3_CODE_OBJECT hook is compiled out of release builds of libckr.so: this probe runs against its own library
    from checkers_mcts_amd import build as ckbuild
    subprocess.check_call([ckbuild.HIPCC] + ckbuild.FLAGS + ["-DCKR_EXPERIMENTS"] + ckbuild.sources() + ["-o", os.path.join(OUT, "libckr_experiments.so")])
    slp = compile_asm("slp", [])
    compile_asm("noslp", ["-fno-slp-vectorize"])
    compile_asm("slp_waitcnt_forcezero", ["-mllvm", "-amdgpu-waitcnt-forcezero"])
    lines = open(slp).readlines()
    cl = clusters(lines)
    whole = (cl[0][0], cl[-1][1])
    meta = {"clusters": [[a + 1, b + 1, sum(1 for ln in lines[a:b + 1] if PK.match(ln))] for a, b in cl]}
    variants = {
        "slp_nop1_after_pk": dict(where=whole, after="s_nop 1"),
        "slp_nop7_after_pk": dict(where=whole, after="s_nop 7"),
        "slp_nop1_before_pk": dict(where=whole, before="s_nop 1"),
        "slp_lgkm0_before_pk": dict(where=whole, before="s_waitcnt lgkmcnt(0)"),
        "slp_vm0_before_pk": dict(where=whole, before="s_waitcnt vmcnt(0)"),
        "slp_nop0_after_sdwa": dict(where=whole, after="s_nop 0", match=re.compile(r"^\s*v_cvt_f32_f16_sdwa\b")),
    }
    for k, c in enumerate(cl):
        variants["slp_nop7_after_pk_cluster%d" % k] = dict(where=c, after="s_nop 7")
    # second round: WHICH counter, and WHERE.  head = from the first packed operation of the policy-head clusters to the end
    # of the kernel; body = everything before it
    INSTR = re.compile(r"^\s*(v_|ds_|global_|buffer_|s_(?!waitcnt|nop|endpgm|cbranch|branch|barrier))")
    VALU, DS, VMEM = re.compile(r"^\s*v_"), re.compile(r"^\s*ds_"), re.compile(r"^\s*(global_|buffer_)")
    head, body = (cl[-2][0] - 40, cl[-1][1]), (0, cl[-2][0] - 41)
    variants.update({
        "slp2_head_all_counters_before_every_instr": dict(where=head, before="s_waitcnt vmcnt(0) lgkmcnt(0)", match=INSTR),
        "slp2_body_all_counters_before_every_instr": dict(where=body, before="s_waitcnt vmcnt(0) lgkmcnt(0)", match=INSTR),
        "slp2_head_lgkm0_after_ds": dict(where=head, after="s_waitcnt lgkmcnt(0)", match=DS),
        "slp2_head_lgkm0_before_valu": dict(where=head, before="s_waitcnt lgkmcnt(0)", match=VALU),
        "slp2_head_vm0_before_every_instr": dict(where=head, before="s_waitcnt vmcnt(0)", match=INSTR),
        "slp2_head_vm0_after_vmem": dict(where=head, after="s_waitcnt vmcnt(0)", match=VMEM),
        "slp2_head_nop3_before_valu": dict(where=head, before="s_nop 3", match=VALU),
    })
    for name, v in variants.items():
        with open(os.path.join(OUT, name + ".s"), "w") as f:
            f.writelines(edit(lines, **v))
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith(".s"):
            assemble(os.path.join(OUT, fn))
    json.dump(meta, open(os.path.join(OUT, "meta.json"), "w"))
    print("built", sorted(f for f in os.listdir(OUT) if f.endswith(".co")), meta)


CHECK = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import copy, numpy as np, torch
from checkers_mcts_amd import net as N, rules
from checkers_mcts_amd.fused import FusedEvaluator
from test_rules_gpu import random_boards
S = 4096
m = N.PolicyValueNet(128).keras_init(3).perturb_bn(7).eval().cuda()
x = rules.features(rules.boards_to_device(random_boards(S, 4242))).contiguous()
fe = FusedEvaluator(m, S, mode="f16x3", debug_outputs=True)
p, v = fe.forward_features(x)
torch.cuda.synchronize()
with torch.no_grad():
    mm = copy.deepcopy(m).double()
    pt, vt = mm(x.permute(0, 3, 1, 2).double())
    ypol = fe.nets[0]["y_pol"].double() / fe.nets[0]["xs_pol"]                       # the kernel's own policy-conv-1 activations
    w = mm.pol2["conv"].weight.reshape(8, 128); b = mm.pol2["conv"].bias
    bn = mm.pol2["bn"]; sc = bn.weight / torch.sqrt(bn.running_var + bn.eps); sh = bn.bias - bn.running_mean * sc
    ref = sc * torch.relu(ypol.reshape(S, 64, 128) @ w.T + b) + sh   # what head_1x1<8> must produce from them
    got = fe.nets[0]["pol_feat"].double().reshape(S, 64, 8)
err = (p.double() - pt).abs().max(1).values.cpu().numpy()
d = (got - ref).abs().cpu().numpy()
bad = d > 1e-4
rows = np.where(bad.any((1, 2)))[0]
out = dict(p_rows_above_1e6=int((err > 1e-6).sum()), p_max_err=float(err.max()), v_max_err=float((v.double() - vt).abs().max()),
           head_rows_wrong=int(len(rows)), head_wrong_by_channel=bad.sum((0, 1)).tolist(),
           head_wrong_by_lane16=[int(bad[:, 16 * q:16 * q + 16].sum()) for q in range(4)],
           head_wrong_rows_mod2=np.bincount(rows % 2, minlength=2).tolist(), first_rows=rows[:8].tolist())
print(json.dumps(out))
'''


def run():
    names = ["linked"] + sorted(f[:-3] for f in os.listdir(OUT) if f.endswith(".co"))
    for name in names:
        env = dict(os.environ, CKR_LIB_PATH=os.path.join(OUT, "libckr_experiments.so"))
        env.pop("CKR_X3_CODE_OBJECT", None)
        if name != "linked":
            env["CKR_X3_CODE_OBJECT"] = os.path.join(OUT, name + ".co")
        r = subprocess.run([sys.executable, "-c", CHECK, ROOT], env=env, capture_output=True, text=True)
        line = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else json.dumps({"error": r.stderr[-400:]})
        print(json.dumps({"variant": name, **json.loads(line)}), flush=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
