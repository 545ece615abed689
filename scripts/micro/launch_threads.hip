// Calibration: what one kernel launch costs the HOST when T threads launch at once, each on a
// non-blocking stream of its own, all on device 0 -- the one-GPU rehearsal of a Manager of T
// shards whose launches come from T host threads (mrx_api.cpp, shard workers).  Kernel arguments
// of 408 bytes like the renderer's.
//   free     every thread launches at its own pace (bursts of 40, then a stream sync)
//   lockstep a round starts for all threads at once and ends when the last launch returned
//            (what mrx_step does); reported: us per round
//   api      <<<>>> (hipLaunchKernel: function looked up per launch) or hipModuleLaunchKernel
//            with the hipFunction_t resolved once
//   kernel   nop (the queue never backs up) or a ~12 us fill (a backlog as in the renderer)
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { unsigned w[100]; };
__global__ void work(Args a, uint4 *p, unsigned n)
{
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = make_uint4(a.w[7], i, i, i);
}

static double nowUs()
{
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const int burst = 40, reps = 50;
    uint4 *buf = nullptr;
    const unsigned nFill = 4u << 20;                          // 64 MiB per launch: ~12 us
    CK(hipMalloc(&buf, (size_t)nFill * 16 * 8));
    hipFunction_t fn = nullptr;
    CK(hipGetFuncBySymbol(&fn, reinterpret_cast<const void *>(&work)));
    for (int heavy = 0; heavy < 2; ++heavy)
        for (int api = 0; api < 2; ++api)
            for (int lock = 0; lock < 2; ++lock)
                for (int T : {1, 2, 4, 8}) {
                    std::vector<hipStream_t> streams(T);
                    for (auto &s : streams) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
                    std::vector<double> us(T, 0.0);
                    std::atomic<int> round { 0 }, arrived { 0 };
                    std::vector<std::thread> th;
                    const unsigned n = heavy ? nFill : 1024u * 256u;
                    auto launch = [&](int t) {
                        Args a {};
                        uint4 *p = buf + (size_t)t * nFill;
                        if (api == 0) {
                            work<<<n / 256, 256, 0, streams[t]>>>(a, p, n);
                        } else {
                            unsigned nn = n;
                            void *args[] = { &a, &p, &nn };
                            CK(hipModuleLaunchKernel(fn, n / 256, 1, 1, 256, 1, 1, 0, streams[t], args, nullptr));
                        }
                    };
                    double roundUs = 0;
                    const int rounds = reps * burst;
                    for (int t = 1; t < T; ++t)
                        th.emplace_back([&, t] {
                            CK(hipSetDevice(0));
                            if (!lock) {
                                double tot = 0;
                                for (int r = 0; r < reps; ++r) {
                                    const double t0 = nowUs();
                                    for (int i = 0; i < burst; ++i) launch(t);
                                    tot += nowUs() - t0;
                                    CK(hipStreamSynchronize(streams[t]));
                                }
                                us[t] = tot / (reps * burst);
                                return;
                            }
                            for (int r = 1; r <= rounds; ++r) {
                                while (round.load(std::memory_order_acquire) < r) __builtin_ia32_pause();
                                launch(t);
                                arrived.fetch_add(1, std::memory_order_release);
                                if (r % burst == 0) CK(hipStreamSynchronize(streams[t]));
                            }
                        });
                    if (!lock) {
                        double tot = 0;
                        for (int r = 0; r < reps; ++r) {
                            const double t0 = nowUs();
                            for (int i = 0; i < burst; ++i) launch(0);
                            tot += nowUs() - t0;
                            CK(hipStreamSynchronize(streams[0]));
                        }
                        us[0] = tot / (reps * burst);
                    } else {
                        for (int r = 1; r <= rounds; ++r) {
                            const double t0 = nowUs();
                            round.store(r, std::memory_order_release);
                            launch(0);
                            while (arrived.load(std::memory_order_acquire) < (T - 1) * r) __builtin_ia32_pause();
                            roundUs += nowUs() - t0;
                            if (r % burst == 0) {
                                CK(hipStreamSynchronize(streams[0]));
                                // (the workers sync their own streams; give them the time)
                                std::this_thread::sleep_for(std::chrono::microseconds(heavy ? 800 : 100));
                            }
                        }
                    }
                    for (auto &x : th) x.join();
                    CK(hipDeviceSynchronize());
                    double worst = 0, sum = 0;
                    for (double u : us) { worst = u > worst ? u : worst; sum += u; }
                    if (!lock)
                        printf("%-5s %-8s free     T=%d  host us per launch: avg %.2f worst thread %.2f\n",
                               heavy ? "fill" : "nop", api ? "module" : "<<<>>>", T, sum / T, worst);
                    else
                        printf("%-5s %-8s lockstep T=%d  host us per round (all T launches returned): %.2f\n",
                               heavy ? "fill" : "nop", api ? "module" : "<<<>>>", T, roundUs / rounds);
                    for (auto &s : streams) CK(hipStreamDestroy(s));
                }
    return 0;
}
