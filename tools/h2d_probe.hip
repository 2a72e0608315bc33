// Developer probe: how to move a few megabytes of PAGEABLE host memory to the device fastest (the .bin payload of an
// 8192^2 image is 5.5 MB): hipMemcpy as is, one thread through a pinned buffer, N threads through pinned buffers.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    const size_t bytes = argc > 1 ? (size_t)atol(argv[1]) : 5563479;
    std::vector<unsigned char> host(bytes, 7);
    void *dev, *pin;
    (void)hipMalloc(&dev, bytes);
    (void)hipHostMalloc(&pin, bytes, hipHostMallocDefault);
    hipStream_t st[8];
    for (auto &s : st)
        (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    for (int rep = 0; rep < 3; ++rep) {
        double t0 = now_ms();
        (void)hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice);
        double t1 = now_ms();
        std::memcpy(pin, host.data(), bytes);
        double t1b = now_ms();
        (void)hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, st[0]);
        (void)hipStreamSynchronize(st[0]);
        double t2 = now_ms();
        printf("hipMemcpy pageable %.3f ms | 1 thread: memcpy %.3f + DMA %.3f = %.3f ms", t1 - t0, t1b - t1, t2 - t1b, t2 - t1);
        for (int n : {2, 4, 8}) {
            double t3 = now_ms();
            std::vector<std::thread> th;
            for (int t = 0; t < n; ++t)
                th.emplace_back([&, t] {
                    const size_t a = bytes * t / n, b = bytes * (t + 1) / n;
                    std::memcpy((char *)pin + a, host.data() + a, b - a);
                    (void)hipMemcpyAsync((char *)dev + a, (char *)pin + a, b - a, hipMemcpyHostToDevice, st[t]);
                    (void)hipStreamSynchronize(st[t]);
                });
            for (auto &x : th)
                x.join();
            printf(" | %d threads %.3f ms", n, now_ms() - t3);
        }
        printf("\n");
    }
    return 0;
}
