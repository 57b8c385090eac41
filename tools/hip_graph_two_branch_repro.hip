// Standalone stress for hipGraphLaunch with TWO-BRANCH graphs under repeated re-capture -- no product code.
//
//     hipcc --offload-arch=gfx950 -O2 tools/hip_graph_two_branch_repro.hip -o /tmp/two_branch_repro
//     timeout 600 /tmp/two_branch_repro [runners=3] [rounds=400] [replays per capture=200]
//
// What the product did when the fault of profiles/r04_arena_streams_fault.txt appeared (SIGSEGV in hip::Graph::UpdateStreams <-
// hip::GraphExec::Run <- hipGraphLaunch, ROCm 7.2 runtime as shipped with PyTorch 2.10+rocm7.0), reduced to HIP calls:
//   * several "runners", each with a stream of its own and a step graph captured from that stream;
//   * every step graph has two branches: kernel A on the capture stream; an event recorded there makes a SIDE stream wait (fork),
//     kernel B runs on the side stream, a second event brings it back (join), kernel C follows on the capture stream;
//   * the runners replay their graphs round-robin (the host never waits for one before launching the next);
//   * every so often a runner destroys its graph + executable and captures a new one (the product re-captured ~10 times per
//     engine through the tail of a run) while the OTHER runners' replays are still in flight.
// The side stream is created once per runner and re-used by every capture of that runner, as torch's pool streams were.
// A run that completes prints "ok"; a fault shows as the process dying inside hipGraphLaunch (run it under rocgdb -batch -ex run -ex bt).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_work(float* p, int n, float a) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float v = p[i]; for (int k = 0; k < 64; ++k) v = v * a + 1.0f; p[i] = v; }
}

struct Runner {
    hipStream_t main = nullptr, side = nullptr;
    hipEvent_t fork = nullptr, join = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    float *a = nullptr, *b = nullptr;
    int n = 0, captures = 0;

    void capture() {
        if (exec) { CHECK(hipGraphExecDestroy(exec)); exec = nullptr; }
        if (graph) { CHECK(hipGraphDestroy(graph)); graph = nullptr; }
        CHECK(hipStreamBeginCapture(main, hipStreamCaptureModeThreadLocal));
        hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, main, a, n, 0.5f);          // A
        CHECK(hipEventRecord(fork, main));
        CHECK(hipStreamWaitEvent(side, fork, 0));                                                    // fork
        hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, side, b, n, 0.25f);         // B, second branch
        hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, main, a, n, 0.75f);         // A', first branch
        CHECK(hipEventRecord(join, side));
        CHECK(hipStreamWaitEvent(main, join, 0));                                                    // join
        hipLaunchKernelGGL(k_work, dim3((n + 255) / 256), dim3(256), 0, main, a, n, 0.5f);          // C
        CHECK(hipStreamEndCapture(main, &graph));
        CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        ++captures;
    }
};

int main(int argc, char** argv) {
    const int runners = argc > 1 ? atoi(argv[1]) : 3, rounds = argc > 2 ? atoi(argv[2]) : 400, replays = argc > 3 ? atoi(argv[3]) : 200;
    std::vector<Runner> R((size_t)runners);
    for (int r = 0; r < runners; ++r) {
        Runner& x = R[(size_t)r];
        x.n = 1 << (14 + r % 3);
        CHECK(hipStreamCreateWithFlags(&x.main, hipStreamNonBlocking));
        CHECK(hipStreamCreateWithFlags(&x.side, hipStreamNonBlocking));
        CHECK(hipEventCreateWithFlags(&x.fork, hipEventDisableTiming));
        CHECK(hipEventCreateWithFlags(&x.join, hipEventDisableTiming));
        CHECK(hipMalloc((void**)&x.a, (size_t)x.n * sizeof(float)));
        CHECK(hipMalloc((void**)&x.b, (size_t)x.n * sizeof(float)));
        CHECK(hipMemset(x.a, 0, (size_t)x.n * sizeof(float)));
        CHECK(hipMemset(x.b, 0, (size_t)x.n * sizeof(float)));
        x.capture();
    }
    long launches = 0;
    for (int round = 0; round < rounds; ++round) {
        for (int i = 0; i < replays; ++i)
            for (int r = 0; r < runners; ++r) { CHECK(hipGraphLaunch(R[(size_t)r].exec, R[(size_t)r].main)); ++launches; }
        // one runner re-captures while the others' replays are in flight (it waits for its own stream only, as the product did)
        Runner& x = R[(size_t)(round % runners)];
        CHECK(hipStreamSynchronize(x.main));
        x.capture();
        if (round % 50 == 49) { printf("round %d: %ld launches, %d captures of runner 0\n", round + 1, launches, R[0].captures); fflush(stdout); }
    }
    CHECK(hipDeviceSynchronize());
    printf("ok: %d runners, %ld two-branch graph launches, %d re-captures\n", runners, launches, rounds);
    return 0;
}
