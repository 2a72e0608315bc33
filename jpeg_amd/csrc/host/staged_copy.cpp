// staged_copy.cpp -- copies between pageable host memory and the device at bus speed.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <mutex>
#include <system_error>
#include <thread>

#include "../host_common.h"

namespace {

// ---- host <-> device copies of pageable memory ------------------------------------------
// hipMemcpy on pageable memory moves about 5 GB/s on the box (8192^2: 40 ms each way, far
// more than every kernel together).  Four worker threads, each with two pinned 4 MiB staging
// buffers and its own stream, copy row ranges instead: the CPU fills (or drains) one buffer
// while the DMA engine moves the other.  There is one pool per device (created at its first
// use, kept for the life of the process): callers on different GPUs neither serialise on one
// another nor recreate each other's streams; concurrent callers on ONE device take turns.
class StagedCopier {
public:
    // `rows` rows of `width` bytes between a dense device buffer and host rows `hpitch` apart
    int copy(void *dev, uint8_t *host, size_t hpitch, size_t width, size_t rows, bool to_device)
    {
        const hipMemcpyKind kind = to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost;
        if (rows == 0 || width == 0)
            return 0;
        if (width * rows < (1u << 20) || width > kChunk) { // small, or rows wider than a staging buffer
            if (to_device)
                JPEGAMD_HIPCHK(hipMemcpy2D(dev, width, host, hpitch, width, rows, kind));
            else
                JPEGAMD_HIPCHK(hipMemcpy2D(host, hpitch, dev, width, width, rows, kind));
            return 0;
        }
        std::lock_guard<std::mutex> lock(mu_);
        int device = 0;
        JPEGAMD_HIPCHK(hipGetDevice(&device));
        if (prepare(device))
            return 1;
        hipError_t err[kThreads];
        std::thread th[kThreads];
        for (int t = 0; t < kThreads; ++t) {
            const size_t r0 = rows * t / kThreads, r1 = rows * (t + 1) / kThreads;
            auto part = [=, &err] {
                err[t] = worker(t, device, static_cast<uint8_t *>(dev) + r0 * width, host + r0 * hpitch, hpitch, width,
                                r1 - r0, to_device);
            };
            try {
                th[t] = std::thread(part);
            } catch (const std::system_error &) { // no thread to be had: this part runs here
                part();
            }
        }
        for (auto &x : th)
            if (x.joinable())
                x.join();
        for (int t = 0; t < kThreads; ++t)
            if (err[t] != hipSuccess)
                return jpegamd_fail("staged copy: %s", hipGetErrorString(err[t]));
        return 0;
    }

    // creates the staging buffers, streams and events for the current device ahead of the first copy
    int warm()
    {
        std::lock_guard<std::mutex> lock(mu_);
        int device = 0;
        JPEGAMD_HIPCHK(hipGetDevice(&device));
        return prepare(device);
    }

private:
#ifndef JPEGAMD_COPY_THREADS
#define JPEGAMD_COPY_THREADS 4
#endif
#ifndef JPEGAMD_COPY_CHUNK_MIB
#define JPEGAMD_COPY_CHUNK_MIB 4
#endif
    static constexpr int kThreads = JPEGAMD_COPY_THREADS;
    static constexpr size_t kChunk = (size_t)JPEGAMD_COPY_CHUNK_MIB << 20;
    std::mutex mu_;
    int device_ = -1;
    void *pin_[kThreads][2] = {};
    hipStream_t stream_[kThreads] = {};
    hipEvent_t done_[kThreads][2] = {};

    int prepare(int device)
    {
        if (device_ == device)
            return 0;
        device_ = -1; // until everything below has succeeded the copier belongs to no device
        for (int t = 0; t < kThreads; ++t) { // first use, or another device: (re)create streams and events
            if (stream_[t])
                (void)hipStreamDestroy(stream_[t]);
            for (int b = 0; b < 2; ++b)
                if (done_[t][b])
                    (void)hipEventDestroy(done_[t][b]);
            stream_[t] = nullptr, done_[t][0] = done_[t][1] = nullptr;
            for (int b = 0; b < 2; ++b)
                if (!pin_[t][b])
                    JPEGAMD_HIPCHK(hipHostMalloc(&pin_[t][b], kChunk, hipHostMallocPortable));
            JPEGAMD_HIPCHK(hipStreamCreateWithFlags(&stream_[t], hipStreamNonBlocking));
            JPEGAMD_HIPCHK(hipEventCreateWithFlags(&done_[t][0], hipEventDisableTiming));
            JPEGAMD_HIPCHK(hipEventCreateWithFlags(&done_[t][1], hipEventDisableTiming));
        }
        device_ = device;
        return 0;
    }

    hipError_t worker(int t, int device, uint8_t *dev, uint8_t *host, size_t hpitch, size_t width, size_t rows,
                      bool to_device)
    {
        hipError_t e = hipSetDevice(device); // the current device is per thread
        const size_t per = std::max<size_t>(1, kChunk / width); // rows per chunk
        const size_t chunks = (rows + per - 1) / per;
        auto span = [&](size_t i) { return std::min(per, rows - i * per); };
        auto host_rows = [&](size_t i, int slot, bool fill) { // pageable rows <-> packed pinned rows
            uint8_t *pin = static_cast<uint8_t *>(pin_[t][slot]);
            for (size_t r = 0; r < span(i); ++r) {
                uint8_t *h = host + (i * per + r) * hpitch;
                if (fill)
                    std::memcpy(pin + r * width, h, width);
                else
                    std::memcpy(h, pin + r * width, width);
            }
        };
        for (size_t i = 0; i < chunks && e == hipSuccess; ++i) {
            const int slot = (int)(i & 1);
            if (to_device) {
                if (i >= 2)
                    e = hipEventSynchronize(done_[t][slot]); // the DMA that last read this buffer
                host_rows(i, slot, true);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(dev + i * per * width, pin_[t][slot], span(i) * width, hipMemcpyHostToDevice,
                                       stream_[t]);
            } else {
                e = hipMemcpyAsync(pin_[t][slot], dev + i * per * width, span(i) * width, hipMemcpyDeviceToHost,
                                   stream_[t]);
            }
            if (e == hipSuccess)
                e = hipEventRecord(done_[t][slot], stream_[t]);
            if (!to_device && i >= 1 && e == hipSuccess) { // drain the previous chunk while this one moves
                e = hipEventSynchronize(done_[t][slot ^ 1]);
                host_rows(i - 1, slot ^ 1, false);
            }
        }
        if (e == hipSuccess)
            e = hipStreamSynchronize(stream_[t]);
        if (!to_device && e == hipSuccess && chunks)
            host_rows(chunks - 1, (int)((chunks - 1) & 1), false);
        return e;
    }
};

constexpr int kMaxDevices = kJpegamdMaxDevices; // host_common.h
StagedCopier g_copiers[kMaxDevices];

StagedCopier *copier_of_current_device()
{
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= kMaxDevices) {
        (void)jpegamd_fail("staged copy: no current device, or a device index beyond %d", kMaxDevices - 1);
        return nullptr;
    }
    return &g_copiers[device];
}

} // namespace

int jpegamd_staged_copy(void *dev, uint8_t *host, size_t hpitch, size_t width, size_t rows, bool to_device)
{
    StagedCopier *c = copier_of_current_device();
    return c ? c->copy(dev, host, hpitch, width, rows, to_device) : 1;
}

int jpegamd_staged_copy_warm()
{
    StagedCopier *c = copier_of_current_device();
    return c ? c->warm() : 1;
}

// A few megabytes (a .bin's payload section is 5.5 MB at 8192^2): ONE thread through a pinned block of the range's
// size, in 1 MiB pieces -- the DMA engine moves piece k while the CPU copies piece k + 1 (every piece has its own
// place in the block, so nothing waits for a buffer).  tools/h2d_probe.hip: 0.19 ms for 5.5 MB against 1.2-1.6 ms
// for hipMemcpy on the pageable range and 0.3-0.55 ms through the four worker threads above, whose start alone
// costs 0.1 ms.
namespace {
struct MidCopier {
    static constexpr size_t kPiece = (size_t)1 << 20, kMax = (size_t)16 << 20;
    void *pin = nullptr;
    size_t cap = 0;
    hipEvent_t ev[kMax / kPiece] = {};
    ~MidCopier()
    {
        if (pin)
            (void)hipHostFree(pin);
        for (auto e : ev)
            if (e)
                (void)hipEventDestroy(e);
    }
    int copy(uint8_t *dev, uint8_t *host, size_t bytes, bool to_device)
    {
        if (cap < bytes) {
            if (pin)
                (void)hipHostFree(pin);
            pin = nullptr, cap = 0;
            const size_t want = std::min(kMax, bytes + bytes / 4);
            JPEGAMD_HIPCHK(hipHostMalloc(&pin, want, hipHostMallocDefault));
            cap = want;
        }
        uint8_t *p = static_cast<uint8_t *>(pin);
        const size_t pieces = (bytes + kPiece - 1) / kPiece;
        if (to_device) {
            for (size_t i = 0; i < pieces; ++i) {
                const size_t o = i * kPiece, n = std::min(kPiece, bytes - o);
                std::memcpy(p + o, host + o, n);
                JPEGAMD_HIPCHK(hipMemcpyAsync(dev + o, p + o, n, hipMemcpyHostToDevice, nullptr));
            }
            JPEGAMD_HIPCHK(hipStreamSynchronize(nullptr)); // the block is free again when this returns
            return 0;
        }
        for (size_t i = 0; i < pieces; ++i) {
            const size_t o = i * kPiece, n = std::min(kPiece, bytes - o);
            if (!ev[i])
                JPEGAMD_HIPCHK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
            JPEGAMD_HIPCHK(hipMemcpyAsync(p + o, dev + o, n, hipMemcpyDeviceToHost, nullptr));
            JPEGAMD_HIPCHK(hipEventRecord(ev[i], nullptr));
        }
        for (size_t i = 0; i < pieces; ++i) {
            const size_t o = i * kPiece, n = std::min(kPiece, bytes - o);
            JPEGAMD_HIPCHK(hipEventSynchronize(ev[i]));
            std::memcpy(host + o, p + o, n);
        }
        return 0;
    }
};
} // namespace

// one contiguous range: small ones directly, a few megabytes through the calling thread's pinned block, large ones
// as whole 4 KiB rows through the staged copier (the tail directly)
int jpegamd_staged_copy_linear(void *dev, void *host, size_t bytes, bool to_device)
{
    if (bytes >= ((size_t)64 << 10) && bytes <= MidCopier::kMax) {
        static thread_local MidCopier mid; // (events and the block belong to the device that is current at first use:
        int device = 0;                    //  another device gets the general path below)
        static thread_local int mid_device = -1;
        JPEGAMD_HIPCHK(hipGetDevice(&device));
        if (mid_device < 0)
            mid_device = device;
        if (mid_device == device)
            return mid.copy(static_cast<uint8_t *>(dev), static_cast<uint8_t *>(host), bytes, to_device);
    }
    const size_t tail = bytes % 4096, head = bytes - tail;
    if (head && jpegamd_staged_copy(dev, static_cast<uint8_t *>(host), 4096, 4096, head / 4096, to_device))
        return 1;
    if (tail) {
        uint8_t *d = static_cast<uint8_t *>(dev) + head, *h = static_cast<uint8_t *>(host) + head;
        if (to_device)
            JPEGAMD_HIPCHK(hipMemcpy(d, h, tail, hipMemcpyHostToDevice));
        else
            JPEGAMD_HIPCHK(hipMemcpy(h, d, tail, hipMemcpyDeviceToHost));
    }
    return 0;
}
