#!/usr/bin/env python3
"""Training-step throughput: the hand-written HIP step (train_hip.HipTrainStep) and the PyTorch
autograd / MIOpen / fused-Adam step on the same network and batch, each captured in a HIP graph.  CKR_TRAIN_PIPE = bf16x6 (default) | f32
selects the matrix pipe of the conv GEMMs.  One JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from checkers_mcts_amd import net as N, train as T
from checkers_mcts_amd.train_hip import HipTrainStep

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda")


def graph_of(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    return g


HOST = []


def timeit(g, reps=50):
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    for _ in range(reps):
        g.replay()
    HOST.append((time.perf_counter() - t0) / reps)                # host time to ISSUE one replay (the queue runs ahead if this is smaller)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    torch.manual_seed(0)
    x = (torch.rand(B, 8, 8, 14, device=dev) < 0.2).float().contiguous()
    pi = torch.softmax(torch.randn(B, 512, device=dev), 1).contiguous()
    tv = (torch.rand(B, device=dev) * 2 - 1).contiguous()
    lr = torch.tensor(1e-3, device=dev)
    acc = torch.zeros(3, dtype=torch.float64, device=dev)
    net = N.PolicyValueNet(128).keras_init(0).float().to(dev)
    hs = HipTrainStep(net, B, 1e-3, 1e-3)
    t_hip = timeit(graph_of(lambda: hs.step(x, pi, tv, lr, acc, B)))
    ref = N.PolicyValueNet(128).keras_init(0).float().to(dev).to(memory_format=torch.channels_last).train()
    ref.conv_reg = ref.dense_reg = 1e-3; ref.policy_loss_weight = ref.value_loss_weight = 1.0
    opt = torch.optim.Adam(ref.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-7, fused=True, capturable=True)

    def torch_step():
        opt.zero_grad(set_to_none=False)
        loss, ce, mse = T.losses(ref, x, pi, tv, None, with_penalty=False)
        loss.backward()
        T.add_l2_gradients(ref)
        opt.step()
    t_torch = timeit(graph_of(torch_step))
    # the conv layers' GEMMs alone: forward + data gradient + weight gradient of the seven 128 -> 128 layers and the first layer
    P = 64 * B
    flops = 2.0 * P * 128 * (3 * 7 * 1152 + 2 * 126)
    out = dict(batch=B, hip_host_issue_ms_per_replay=HOST[0] * 1e3, torch_host_issue_ms_per_replay=HOST[1] * 1e3, hip_ms_per_step=t_hip * 1e3, hip_samples_per_s=B / t_hip, torch_ms_per_step=t_torch * 1e3,
               torch_samples_per_s=B / t_torch, speedup=t_torch / t_hip, conv_gemm_flops_per_step=flops,
               conv_gemm_tflops_if_whole_step=flops / t_hip / 1e12, fp32_matrix_peak_tflops=157.3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
