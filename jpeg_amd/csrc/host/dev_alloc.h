// dev_alloc.h -- device buffers of the host-pointer pipeline (pipeline.cpp, entropy_kernels.hip).
// Without a context a buffer is allocated and freed per call, like the reference's
// kernel.cu:97-104.  Inside a jpegamd_context_* call the calling thread has an arena: the n-th
// buffer a call asks for is the n-th slot of the context, reused while it is large enough and
// replaced by a larger one otherwise, and nothing is freed before jpegamd_context_destroy.
#ifndef JPEGAMD_DEV_ALLOC_H
#define JPEGAMD_DEV_ALLOC_H
#include <hip/hip_runtime.h>

#include <cstddef>
#include <vector>

struct JpegamdArena {
    struct Slot {
        void *p;
        size_t n;
    };
    std::vector<Slot> slots;
    size_t next = 0;
    void rewind() { next = 0; }
    void *take(size_t n)
    {
        if (next == slots.size())
            slots.push_back(Slot{nullptr, 0});
        Slot &s = slots[next++];
        if (s.n < n) {
            if (s.p)
                (void)hipFree(s.p);
            s.p = nullptr, s.n = 0;
            const size_t want = n + n / 8; // data-dependent sizes (pairs, payloads) wobble from image to image
            if (hipMalloc(&s.p, want) != hipSuccess)
                return nullptr;
            s.n = want;
        }
        return s.p;
    }
    void release()
    {
        for (Slot &s : slots)
            if (s.p)
                (void)hipFree(s.p);
        slots.clear();
        next = 0;
    }
    ~JpegamdArena() { release(); }
};

// the arena of the context call running on this thread, or null
__attribute__((visibility("hidden"))) JpegamdArena *&jpegamd_thread_arena();

struct JpegamdDevMem {
    void *p = nullptr;
    bool owned = true;
    JpegamdDevMem() = default;
    JpegamdDevMem(const JpegamdDevMem &) = delete;
    JpegamdDevMem &operator=(const JpegamdDevMem &) = delete;
    ~JpegamdDevMem() { release(); }
    int alloc(size_t n)
    {
        release();
        n = n ? n : 16;
        if (JpegamdArena *a = jpegamd_thread_arena()) {
            p = a->take(n);
            owned = false;
            return p ? 0 : 1;
        }
        owned = true;
        return hipMalloc(&p, n) == hipSuccess ? 0 : 1;
    }
    void release()
    {
        if (p && owned)
            (void)hipFree(p);
        p = nullptr;
    }
    template <class T> T *as() { return static_cast<T *>(p); }
};
#endif
