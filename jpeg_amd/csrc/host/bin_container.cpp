// bin_container.cpp -- the reference's entropy stage and .bin container on the host,
// bit-compatible with encode_decode.cpp:375-482 (frequency dictionary, Huffman
// tree, codes, encode/decode) and :1093-1302 (bit packing, saveEncodedData,
// loadEncodedData).  Part of libjpeg_amd.so.
//
// What makes a .bin reproducible is not the Huffman algorithm but two pieces of
// libstdc++ behaviour the reference inherits:
//   * the dictionary is a std::unordered_map<int,int> filled by freq[s]++ over the
//     mixed (value, run) symbol stream (:375-383) and written in the container's
//     iteration order (:1201-1217).  That order is a function of the sequence of
//     *insertions* only, so a flat histogram plus the list of first occurrences,
//     replayed into a real unordered_map, gives the same container;
//   * equal frequencies leave the tree shape to std::priority_queue's sift order
//     (:403-442: leaves pushed in (frequency, key) order, internal nodes compare
//     by frequency alone).  The same adaptor with the same comparator and the same
//     push/pop sequence is used here, over node indices instead of heap nodes.
// Everything else is done differently from the reference for speed: no '0'/'1'
// std::string (:458-466), codes live in flat tables indexed by symbol and bits are
// packed MSB-first as they are produced (:1093-1127 give the byte layout).
#include <algorithm>
#include <exception>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <queue>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/jpeg_amd.h"
#include "../host_common.h"
#include "entropy_internal.h"

namespace {

int fail(const char *text) { return jpegamd_fail("%s", text); }

constexpr int kInternal = 99999; // the reference's marker for internal nodes (:424)

struct Tree {
    // node i: leaf (left < 0) carrying `value`, or internal with two children
    std::vector<int> value, left, right;
    std::vector<long long> freq;
    int root = -1;
    int add(int v, long long f, int l, int r)
    {
        value.push_back(v), freq.push_back(f), left.push_back(l), right.push_back(r);
        return (int)value.size() - 1;
    }
    bool leaf(int n) const { return left[n] < 0; }
};

// build_huffman_tree (:403-442) from (key, frequency) pairs in container order
void build_tree(const std::vector<std::pair<int, int>> &dict, Tree &t)
{
    std::vector<std::pair<int, int>> fv(dict);
    std::sort(fv.begin(), fv.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) {
        return a.second != b.second ? a.second < b.second : a.first < b.first;
    });
    auto after = [&t](int a, int b) { return t.freq[a] > t.freq[b]; }; // Compare, :395-401
    std::priority_queue<int, std::vector<int>, decltype(after)> heap(after);
    for (const auto &p : fv)
        heap.push(t.add(p.first, p.second, -1, -1));
    while (heap.size() > 1) {
        const int l = heap.top();
        heap.pop();
        const int r = heap.top();
        heap.pop();
        heap.push(t.add(kInternal, t.freq[l] + t.freq[r], l, r));
    }
    t.root = heap.empty() ? -1 : heap.top();
}

struct Code {
    std::vector<uint8_t> bits; // one 0/1 per entry: codes can be longer than a machine word
};

// generate_huffman_codes (:444-456): left = 0, right = 1.  A key that were to equal
// the internal marker would get no code there either; such input is rejected earlier.
void assign_codes(const Tree &t, std::unordered_map<int, Code> &codes)
{
    if (t.root < 0)
        return;
    std::vector<std::pair<int, std::vector<uint8_t>>> stack;
    stack.push_back({t.root, {}});
    while (!stack.empty()) {
        auto cur = std::move(stack.back());
        stack.pop_back();
        if (t.leaf(cur.first)) {
            codes[t.value[cur.first]].bits = std::move(cur.second);
            continue;
        }
        std::vector<uint8_t> l = cur.second, r = std::move(cur.second);
        l.push_back(0), r.push_back(1);
        stack.push_back({t.left[cur.first], std::move(l)});
        stack.push_back({t.right[cur.first], std::move(r)});
    }
}

struct BitWriter { // stringToBitstream's layout (:1093-1127): MSB first, last byte zero padded
    uint8_t *p;
    uint64_t acc = 0;
    int n;
    explicit BitWriter(uint8_t *dst, int lead_zero_bits = 0) : p(dst), n(lead_zero_bits) {}
    void put(uint64_t code, int len) // len <= 32 per call; the caller sized the buffer
    {
        acc = (acc << len) | code;
        n += len;
        if (n >= 32) {
            const uint32_t w = __builtin_bswap32((uint32_t)(acc >> (n - 32)));
            std::memcpy(p, &w, 4);
            p += 4;
            n -= 32;
        }
    }
    uint8_t *finish()
    {
        while (n >= 8) {
            *p++ = (uint8_t)(acc >> (n - 8));
            n -= 8;
        }
        if (n > 0)
            *p++ = (uint8_t)(acc << (8 - n));
        n = 0;
        return p;
    }
};

typedef JpegamdPlaneCode Plane;

constexpr int kDense = 1 << 17; // symbols in [-65536, 65535] use flat tables

// worker threads for the two passes over the symbols (host cores are otherwise idle
// while this runs); small planes stay on the calling thread
unsigned plane_threads(size_t n)
{
    unsigned hw = std::thread::hardware_concurrency();
    hw = hw ? hw : 1;
    const size_t want = n / (1u << 19) + 1; // about half a million symbols per thread
    return (unsigned)std::min<size_t>(std::min<size_t>(want, hw), 16);
}

// f(chunk index) on nt threads.  The lambdas allocate (vectors, the window table): an exception
// in a worker must not reach std::terminate, and one on the calling thread must not leave
// joinable threads behind -- every chunk's exception is caught where it happens, all threads are
// joined, and the first one is rethrown on the caller (jpegamd_guarded turns it into a status).
template <class F> void parallel_chunks(unsigned nt, F f)
{
    if (nt <= 1) {
        f(0u);
        return;
    }
    std::vector<std::exception_ptr> err(nt);
    auto guarded = [&](unsigned i) {
        try {
            f(i);
        } catch (...) {
            err[i] = std::current_exception();
        }
    };
    std::vector<std::thread> th;
    unsigned started = 1;
    try {
        th.reserve(nt - 1);
        for (; started < nt; ++started)
            th.emplace_back(guarded, started);
    } catch (...) { // no more threads (or memory for them) to be had: the rest runs here
    }
    guarded(0u);
    for (unsigned i = started; i < nt; ++i)
        guarded(i);
    for (auto &t : th)
        t.join();
    for (auto &e : err)
        if (e)
            std::rethrow_exception(e);
}

// ---------------------------------------------------------------- plane decoder
// huffman_decode (:468-482) walks the tree bit by bit over the whole payload.  Here a 12-bit
// window resolves up to four whole codes per step (the average code is 2-3 bits); a code
// longer than the window continues bit by bit from the node the window reaches, so the
// symbol sequence is exactly the bit-serial walk's.
//
// A plane's bit stream has no restart points, but prefix codes resynchronise: a decoder
// started at a wrong bit position falls into step with the true one after a few codes.  Large
// planes are therefore decoded by several threads.  Pass 1: every thread decodes its chunk
// from the chunk's first bit as if a code began there, only counting, and notes where its
// first kHeads codes begin.  Stitch (sequential, short): from the true end of the previous
// chunk the true decoding is continued code by code until it reaches a position the next
// thread has noted -- from there on that thread's codes ARE the true ones, which gives the
// exact bit position and symbol offset of a region per thread.  Pass 2: every thread decodes
// its region into place.  A chunk that never falls into step (possible, never seen) is
// simply decoded as part of its predecessor's region.
class PlaneDecoder {
  public:
    PlaneDecoder(const Tree &t, const uint8_t *payload, unsigned long long nbits) : t_(t), p_(payload), nbits_(nbits)
    {
        win_.resize(1u << kWin);
        for (int v = 0; v < (1 << kWin); ++v) {
            Win &w = win_[v];
            w = Win{};
            int cur = t.root, used = 0;
            while (used < kWin) {
                cur = ((v >> (kWin - 1 - used)) & 1) ? t.right[cur] : t.left[cur];
                ++used;
                if (t.leaf(cur)) {
                    w.sym[w.count++] = t.value[cur];
                    w.used = (uint8_t)used;
                    cur = t.root;
                    if (w.count == kMaxSym)
                        break;
                }
            }
            w.node = cur;
            if (w.count == 0)
                w.used = kWin;
        }
    }

    // all symbols of the plane; returns their number (more than cap: the buffer was too small,
    // only the first cap were written)
    size_t decode(int32_t *out, size_t cap, unsigned threads)
    {
        const unsigned long long chunk_bits = threads > 1 ? (nbits_ / threads) & ~7ull : 0;
        if (threads <= 1 || chunk_bits < 64)
            return run(0, nbits_, false, out, cap, 0, nullptr).n;
        // pass 1
        struct Chunk {
            std::vector<unsigned long long> heads; // where its first codes begin
            unsigned long long end = 0;            // first code boundary at or after the next chunk's start
            size_t count = 0;
        };
        std::vector<Chunk> ch(threads);
        parallel_chunks(threads, [&](unsigned k) {
            const unsigned long long a = k * chunk_bits, b = k + 1 == threads ? nbits_ : (k + 1) * chunk_bits;
            const Progress r = run(a, b, false, nullptr, 0, 0, &ch[k].heads);
            ch[k].end = r.pos, ch[k].count = r.n;
        });
        // stitch: region k = [start[k], start[k + 1]) holds the symbols first[k] .. first[k + 1] - 1
        std::vector<unsigned long long> start(threads + 1, nbits_);
        std::vector<size_t> first(threads + 1, 0);
        start[0] = 0;
        unsigned long long pos = ch[0].end; // true position, always a code boundary
        size_t n = ch[0].count;             // symbols before pos
        bool stream_done = pos >= nbits_;
        for (unsigned k = 1; k < threads; ++k) {
            const Chunk &c = ch[k];
            const unsigned long long chunk_end = k + 1 == threads ? nbits_ : (k + 1) * chunk_bits;
            size_t i = 0;
            bool in_step = false;
            while (!stream_done) {
                while (i < c.heads.size() && c.heads[i] < pos)
                    ++i;
                if (i < c.heads.size() && c.heads[i] == pos) {
                    in_step = true;
                    break;
                }
                if (i == c.heads.size() || pos >= chunk_end)
                    break; // beyond what the thread noted: treat the chunk as part of this region
                const Progress r = run(pos, nbits_, false, nullptr, 0, 0, nullptr, 1); // one true code
                if (r.n == 0)
                    stream_done = true; // ran out of bits inside a code
                pos = r.pos, n += r.n;
            }
            if (in_step) {
                start[k] = pos, first[k] = n;
                n += c.count - i;
                pos = c.end;
                stream_done = pos >= nbits_;
            } else {
                start[k] = ~0ull; // empty region
                if (!stream_done && pos < chunk_end) { // the rare sequential stretch
                    const Progress r = run(pos, chunk_end, false, nullptr, 0, 0, nullptr);
                    n += r.n, pos = r.pos;
                    stream_done = pos >= nbits_ || r.stalled;
                }
            }
        }
        // empty regions take their successor's start
        first[threads] = n;
        for (unsigned k = threads; k-- > 1;)
            if (start[k] == ~0ull)
                start[k] = start[k + 1], first[k] = first[k + 1];
        if (!out)
            return n;
        // pass 2
        parallel_chunks(threads, [&](unsigned k) {
            if (start[k] < start[k + 1])
                // (cap: never past the region's own symbols -- the window writes four slots at a time)
                run(start[k], start[k + 1], start[k + 1] < nbits_, out, std::min(cap, first[k + 1]), first[k], nullptr);
        });
        return n;
    }

  private:
    static constexpr int kWin = 12, kMaxSym = 4;
    static constexpr size_t kHeads = 512;
    struct Win {
        int32_t sym[kMaxSym]; // values of the whole codes found, in order
        int32_t node;         // count == 0: the inner node reached after kWin bits
        uint8_t count, used;  // whole codes found / bits they take (kWin if count == 0)
    };
    struct Progress {
        unsigned long long pos; // a code boundary (or the end of the stream)
        size_t n;               // symbols found by this call
        bool stalled;           // the stream ended inside a code
    };
    const Tree &t_;
    const uint8_t *p_;
    const unsigned long long nbits_;
    std::vector<Win> win_;

    int bit_at(unsigned long long i) const { return (p_[i >> 3] >> (7 - (i & 7))) & 1; }

    // Decodes codes from the boundary `pos` while pos < stop.  exact: `stop` is a true boundary
    // and no code beyond it may be consumed (the window is used only where it cannot cross it).
    // heads: note where the first kHeads codes begin (those go one code at a time).
    // Symbols go to out[n0 + ...] while that index is below cap (out may be null: counting).
    Progress run(unsigned long long pos, unsigned long long stop, bool exact, int32_t *out, size_t cap, size_t n0,
                 std::vector<unsigned long long> *heads, size_t max_codes = ~(size_t)0) const
    {
        size_t n = 0;
        const bool single = max_codes != ~(size_t)0;
        while (pos < stop && n < max_codes) {
            int cur = t_.root;
            const bool noting = heads && heads->size() < kHeads;
            if (noting)
                heads->push_back(pos);
            if (!noting && !single && pos + 64 <= nbits_ && (!exact || pos + kWin <= stop)) {
                uint64_t bits;
                std::memcpy(&bits, p_ + (pos >> 3), 8);
                bits = __builtin_bswap64(bits) << (pos & 7);
                const Win &w = win_[(size_t)(bits >> (64 - kWin))];
                pos += w.used;
                if (w.count) {
                    const size_t at = n0 + n;
                    if (at + kMaxSym <= cap) { // room for all four slots: no per-symbol test
                        for (int i = 0; i < kMaxSym; ++i)
                            out[at + i] = w.sym[i];
                    } else {
                        for (int i = 0; i < w.count; ++i)
                            if (at + i < cap)
                                out[at + i] = w.sym[i];
                    }
                    n += w.count;
                    continue;
                }
                cur = w.node;
            }
            while (!t_.leaf(cur) && pos < nbits_)
                cur = bit_at(pos++) ? t_.right[cur] : t_.left[cur];
            if (!t_.leaf(cur)) {
                if (noting)
                    heads->pop_back();
                return {pos, n, true}; // ran out of bits inside a code: the reference emits nothing for it either
            }
            if (n0 + n < cap)
                out[n0 + n] = t_.value[cur];
            ++n;
        }
        return {pos, n, false};
    }
};

int encode_plane(const int32_t *sym, size_t n, Plane &pl)
{
    const unsigned nt = plane_threads(n);
    auto lo = [&](unsigned c) { return n * c / nt; };
    auto dense = [](int s) { return s >= -kDense / 2 && s < kDense / 2; };

    // ---- frequency dictionary: counts + first-occurrence order ----------------
    // per chunk in parallel, then merged in chunk order: a symbol's first occurrence
    // overall is its first occurrence in the earliest chunk that has it
    std::vector<std::vector<int>> c_count(nt), c_order(nt);
    std::vector<int> c_bad(nt, 0); // 1: symbol outside the dense window, 2: the internal marker
    parallel_chunks(nt, [&](unsigned c) {
        std::vector<int> &cnt = c_count[c];
        cnt.assign(kDense, 0);
        for (size_t i = lo(c), e = lo(c + 1); i < e; ++i) {
            const int s = sym[i];
            if (!dense(s)) {
                c_bad[c] |= s == kInternal ? 2 : 1;
                continue;
            }
            if (cnt[s + kDense / 2]++ == 0)
                c_order[c].push_back(s);
        }
    });
    int bad = 0;
    for (int b : c_bad)
        bad |= b;
    if (bad & 2)
        return fail("bin_encode: symbol 99999 collides with the reference's internal-node marker");
    std::vector<int> &count = c_count[0];
    std::unordered_map<int, int> sparse; // symbols outside the dense window (never from the kernels)
    std::vector<int> order;
    if (bad) { // rare general case: one serial pass over everything
        count.assign(kDense, 0);
        for (size_t i = 0; i < n; ++i) {
            const int s = sym[i];
            int &c = dense(s) ? count[s + kDense / 2] : sparse[s];
            if (c++ == 0)
                order.push_back(s);
        }
    } else {
        std::vector<uint8_t> seen(kDense, 0);
        for (unsigned c = 0; c < nt; ++c) {
            for (int s : c_order[c])
                if (!seen[s + kDense / 2]) {
                    seen[s + kDense / 2] = 1;
                    order.push_back(s);
                }
            if (c)
                for (int s : c_order[c])
                    count[s + kDense / 2] += c_count[c][s + kDense / 2];
        }
    }
    std::unordered_map<int, int> freq; // the reference's container, same insertion sequence
    for (int s : order)
        freq[s] = dense(s) ? count[s + kDense / 2] : sparse[s];
    pl.dict.assign(freq.begin(), freq.end());

    // ---- tree and codes ---------------------------------------------------------
    Tree t;
    build_tree(pl.dict, t);
    std::unordered_map<int, Code> codes;
    assign_codes(t, codes);

    // ---- payload ----------------------------------------------------------------
    // flat (code, length) tables for the dense window when every code fits 32 bits
    bool fast = sparse.empty();
    unsigned long long total_bits = 0;
    for (const auto &kv : codes) {
        fast = fast && kv.second.bits.size() <= 32;
        total_bits += (unsigned long long)kv.second.bits.size() * (unsigned long long)freq[kv.first];
    }
    pl.payload.assign((size_t)((total_bits + 7) / 8) + 8, 0);
    if (fast) {
        std::vector<uint32_t> cbits(kDense, 0);
        std::vector<uint8_t> clen(kDense, 0);
        for (const auto &kv : codes) {
            uint32_t v = 0;
            for (uint8_t b : kv.second.bits)
                v = (v << 1) | b;
            cbits[kv.first + kDense / 2] = v;
            clen[kv.first + kDense / 2] = (uint8_t)kv.second.bits.size();
        }
        // where each chunk's bits start: its symbols' code lengths, from the chunk's own counts
        std::vector<unsigned long long> start(nt + 1, 0);
        for (unsigned c = 0; c < nt; ++c) {
            unsigned long long bits = 0;
            if (nt == 1) {
                bits = total_bits;
            } else {
                for (int s : c_order[c]) // (c_count[0] was merged into the totals: recount chunk 0 below)
                    bits += c ? (unsigned long long)c_count[c][s + kDense / 2] * clen[s + kDense / 2] : 0;
            }
            start[c + 1] = bits;
        }
        if (nt > 1) { // chunk 0 = the total minus the others
            unsigned long long others = 0;
            for (unsigned c = 1; c < nt; ++c)
                others += start[c + 1];
            start[1] = total_bits - others;
            for (unsigned c = 1; c < nt; ++c)
                start[c + 1] += start[c];
        }
        // every chunk packs into its own buffer, starting inside its first byte; the
        // buffers are then laid down in order, OR-ing the byte two chunks share
        std::vector<std::vector<uint8_t>> part(nt);
        parallel_chunks(nt, [&](unsigned c) {
            const unsigned long long b0 = start[c], b1 = start[c + 1];
            std::vector<uint8_t> &buf = part[c];
            buf.assign((size_t)((b1 + 7) / 8 - b0 / 8) + 8, 0);
            BitWriter w(buf.data(), (int)(b0 & 7));
            for (size_t i = lo(c), e = lo(c + 1); i < e; ++i)
                w.put(cbits[sym[i] + kDense / 2], clen[sym[i] + kDense / 2]);
            buf.resize((size_t)(w.finish() - buf.data()));
        });
        for (unsigned c = 0; c < nt; ++c) {
            if (part[c].empty())
                continue;
            uint8_t *dst = pl.payload.data() + start[c] / 8;
            dst[0] |= part[c][0];
            if (part[c].size() > 1)
                std::memcpy(dst + 1, part[c].data() + 1, part[c].size() - 1);
        }
        pl.payload.resize((size_t)((total_bits + 7) / 8));
    } else {
        BitWriter w(pl.payload.data());
        for (size_t i = 0; i < n; ++i)
            for (uint8_t b : codes[sym[i]].bits)
                w.put(b, 1);
        pl.payload.resize((size_t)(w.finish() - pl.payload.data()));
    }
    if (pl.payload.size() > 0x7FFFFFFFu)
        return fail("bin_encode: payload exceeds the container's int32 length field");
    return 0;
}

void put_i32(uint8_t *&p, int v)
{
    std::memcpy(p, &v, 4); // file.write(reinterpret_cast<char*>(&int), 4): native little endian (:1162)
    p += 4;
}

int get_i32(const uint8_t *p)
{
    int v;
    std::memcpy(&v, p, 4);
    return v;
}

} // namespace

int jpegamd_host_encode_plane(const int32_t *sym, size_t n, JpegamdPlaneCode &pl) { return encode_plane(sym, n, pl); }

int jpegamd_bin_parse(const uint8_t *bin, size_t len, JpegamdBinParts &parts)
{
    if (len < 36)
        return fail("bin_decode: shorter than the 9-int header");
    for (int i = 0; i < 6; ++i)
        parts.dims[i] = get_i32(bin + 4 * i);
    size_t off = 36;
    for (int c = 0; c < 3; ++c) {
        const int l = get_i32(bin + 24 + 4 * c);
        if (l < 0 || (size_t)l > len - off)
            return fail("bin_decode: payload length runs past the end of the file");
        parts.payload[c] = bin + off, parts.len[c] = (size_t)l;
        off += (size_t)l;
    }
    if (len - off < 12)
        return fail("bin_decode: dictionary sizes missing");
    for (int c = 0; c < 3; ++c)
        parts.dsz[c] = get_i32(bin + off + 4 * c);
    off += 12;
    for (int c = 0; c < 3; ++c) {
        if (parts.dsz[c] < 0 || (size_t)parts.dsz[c] > (len - off) / 8)
            return fail("bin_decode: dictionary runs past the end of the file");
        parts.dict[c] = bin + off;
        off += (size_t)parts.dsz[c] * 8;
    }
    return 0;
}

int jpegamd_host_decode_plane(const uint8_t *payload, size_t len, const uint8_t *dict, int dsz, int32_t *sym, size_t cap,
                              size_t *n, unsigned threads)
{
    *n = 0;
    std::unordered_map<int, int> freq; // :1279: insertion in file order
    for (int i = 0; i < dsz; ++i)
        freq[get_i32(dict + 8 * (size_t)i)] = get_i32(dict + 8 * (size_t)i + 4);
    Tree t;
    build_tree(std::vector<std::pair<int, int>>(freq.begin(), freq.end()), t); // decodeChannel, :664
    if (len == 0)
        return 0;
    if (t.root < 0 || t.leaf(t.root))
        return fail("bin_decode: a plane with fewer than two distinct symbols has no decodable code "
                    "(the reference dereferences a null child here, encode_decode.cpp:474)");
    PlaneDecoder dec(t, payload, (unsigned long long)len * 8);
    *n = dec.decode(sym, sym ? cap : 0, threads ? threads : 1);
    return 0;
}

int jpegamd_decode_tables(const uint8_t *dict, int dsz, JpegamdDecodeTables &out)
{
    std::unordered_map<int, int> freq; // :1279: insertion in file order
    for (int i = 0; i < dsz; ++i)
        freq[get_i32(dict + 8 * (size_t)i)] = get_i32(dict + 8 * (size_t)i + 4);
    Tree t;
    build_tree(std::vector<std::pair<int, int>>(freq.begin(), freq.end()), t); // decodeChannel, :664
    if (t.root < 0 || t.leaf(t.root))
        return 2;
    for (size_t i = 0; i < t.value.size(); ++i)
        if (t.leaf((int)i) && (t.value[i] < -32768 || t.value[i] > 32767))
            return 2;
    if (t.value.size() >= (1u << 24))
        return 2;
    out.left.assign(t.left.begin(), t.left.end());
    out.right.assign(t.right.begin(), t.right.end());
    out.value.assign(t.value.begin(), t.value.end());
    out.root = t.root;
    out.window.assign((size_t)1 << kDecodeWindow, 0);
    for (int v = 0; v < (1 << kDecodeWindow); ++v) {
        int cur = t.root, used = 0;
        while (used < kDecodeWindow && !t.leaf(cur)) {
            cur = ((v >> (kDecodeWindow - 1 - used)) & 1) ? t.right[cur] : t.left[cur];
            ++used;
        }
        out.window[v] = t.leaf(cur) ? ((uint32_t)used << 24) | (uint32_t)(t.value[cur] + 32768) : (uint32_t)cur;
    }
    return 0;
}

int jpegamd_codes_from_order(const std::vector<int> &order, const uint32_t *count, JpegamdPlaneCode &pl,
                             std::vector<uint32_t> &cbits, std::vector<uint8_t> &clen, unsigned long long &total_bits)
{
    std::unordered_map<int, int> freq; // the reference's container, same insertion sequence (:375-383)
    for (int s : order) {
        if (s == kInternal)
            return fail("bin_encode: symbol 99999 collides with the reference's internal-node marker");
        freq[s] = (int)count[s + 32768];
    }
    pl.dict.assign(freq.begin(), freq.end());
    Tree t;
    build_tree(pl.dict, t);
    std::unordered_map<int, Code> codes;
    assign_codes(t, codes);
    cbits.assign(65536, 0);
    clen.assign(65536, 0);
    total_bits = 0;
    for (const auto &kv : codes) {
        if (kv.second.bits.size() > 32)
            return 2;
        uint32_t v = 0;
        for (uint8_t b : kv.second.bits)
            v = (v << 1) | b;
        cbits[kv.first + 32768] = v;
        clen[kv.first + 32768] = (uint8_t)kv.second.bits.size();
        total_bits += (unsigned long long)kv.second.bits.size() * count[kv.first + 32768];
    }
    return 0;
}

int jpegamd_bin_assemble(const JpegamdPlaneCode (&pl)[3], int y_rows, int y_cols, int c_rows, int c_cols, uint8_t *out,
                         size_t cap, size_t *out_len)
{
    size_t need = 9 * 4 + 3 * 4;
    for (const Plane &p : pl) {
        if (p.payload.size() > 0x7FFFFFFFu)
            return fail("bin_encode: payload exceeds the container's int32 length field");
        need += p.payload.size() + 8 * p.dict.size();
    }
    *out_len = need;
    if (!out || cap < need)
        return fail("bin_encode: output buffer too small (the size needed is in *out_len)");
    uint8_t *w = out;
    // saveEncodedData, :1161-1217
    put_i32(w, y_rows), put_i32(w, y_cols);
    put_i32(w, c_rows), put_i32(w, c_cols);
    put_i32(w, c_rows), put_i32(w, c_cols);
    for (const Plane &p : pl)
        put_i32(w, (int)p.payload.size());
    for (const Plane &p : pl) {
        if (!p.payload.empty()) // (memcpy from an empty vector's null data() is undefined)
            std::memcpy(w, p.payload.data(), p.payload.size());
        w += p.payload.size();
    }
    for (const Plane &p : pl)
        put_i32(w, (int)p.dict.size());
    for (const Plane &p : pl)
        for (const auto &kv : p.dict)
            put_i32(w, kv.first), put_i32(w, kv.second);
    return 0;
}

extern "C" {

static int bin_encode_impl(const int32_t *sym_y, size_t n_y, const int32_t *sym_cb, size_t n_cb,
                       const int32_t *sym_cr, size_t n_cr, int y_rows, int y_cols, int c_rows, int c_cols,
                       uint8_t *out, size_t cap, size_t *out_len)
{
    if (!out_len || (!sym_y && n_y) || (!sym_cb && n_cb) || (!sym_cr && n_cr))
        return fail("bin_encode: null pointer");
    Plane pl[3];
    if (encode_plane(sym_y, n_y, pl[0]) || encode_plane(sym_cb, n_cb, pl[1]) || encode_plane(sym_cr, n_cr, pl[2]))
        return 1;
    return jpegamd_bin_assemble(pl, y_rows, y_cols, c_rows, c_cols, out, cap, out_len);
}

static int bin_decode_impl(const uint8_t *bin, size_t len, int32_t dims[6],
                       int32_t *sym_y, size_t cap_y, size_t *n_y,
                       int32_t *sym_cb, size_t cap_cb, size_t *n_cb,
                       int32_t *sym_cr, size_t cap_cr, size_t *n_cr)
{
    if (!bin || !dims || !n_y || !n_cb || !n_cr)
        return fail("bin_decode: null pointer");
    *n_y = *n_cb = *n_cr = 0;
    // loadEncodedData, :1236-1296
    if (len < 36)
        return fail("bin_decode: shorter than the 9-int header");
    for (int i = 0; i < 6; ++i)
        dims[i] = get_i32(bin + 4 * i);
    const int lens[3] = {get_i32(bin + 24), get_i32(bin + 28), get_i32(bin + 32)};
    size_t off = 36;
    const uint8_t *payload[3];
    for (int c = 0; c < 3; ++c) {
        if (lens[c] < 0 || (size_t)lens[c] > len - off)
            return fail("bin_decode: payload length runs past the end of the file");
        payload[c] = bin + off;
        off += (size_t)lens[c];
    }
    if (len - off < 12)
        return fail("bin_decode: dictionary sizes missing");
    const int dsz[3] = {get_i32(bin + off), get_i32(bin + off + 4), get_i32(bin + off + 8)};
    off += 12;
    int32_t *outs[3] = {sym_y, sym_cb, sym_cr};
    const size_t caps[3] = {cap_y, cap_cb, cap_cr};
    size_t *ns[3] = {n_y, n_cb, n_cr};
    size_t dict_off[3];
    for (int c = 0; c < 3; ++c) {
        if (dsz[c] < 0 || (size_t)dsz[c] > (len - off) / 8)
            return fail("bin_decode: dictionary runs past the end of the file");
        dict_off[c] = off;
        off += (size_t)dsz[c] * 8;
    }
    // Threads per plane (PlaneDecoder): in proportion to the payloads, 16 in all at most, none
    // extra for less than 64 KiB of payload each (JPEGAMD_DECODE_CHUNK_BYTES overrides that
    // size; the tests use it to cut small files into many chunks).
    unsigned chunk_threads[3] = {1, 1, 1};
    {
        size_t min_chunk = 1u << 16;
        if (const char *e = getenv("JPEGAMD_DECODE_CHUNK_BYTES"))
            min_chunk = (size_t)std::max(8L, atol(e));
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t budget = std::min(hw, 16u);
        const size_t total = (size_t)lens[0] + (size_t)lens[1] + (size_t)lens[2];
        for (int c = 0; c < 3 && total > 0; ++c) {
            const size_t share = std::max<size_t>(1, budget * (size_t)lens[c] / total);
            chunk_threads[c] = (unsigned)std::max<size_t>(1, std::min(share, (size_t)lens[c] / min_chunk));
        }
    }
    // The three planes are independent streams: one thread each when the file is large.
    // Workers only report a status; the message is set on the calling thread.
    enum { kOk = 0, kNoCode = 1, kTooSmall = 2 };
    int status[3] = {kOk, kOk, kOk};
    auto decode_plane = [&](int c) {
        std::unordered_map<int, int> freq; // :1279: insertion in file order
        for (int i = 0; i < dsz[c]; ++i)
            freq[get_i32(bin + dict_off[c] + 8 * (size_t)i)] = get_i32(bin + dict_off[c] + 8 * (size_t)i + 4);
        Tree t;
        build_tree(std::vector<std::pair<int, int>>(freq.begin(), freq.end()), t); // decodeChannel, :664
        size_t n = 0;
        if (lens[c] > 0) {
            if (t.root < 0 || t.leaf(t.root)) {
                status[c] = kNoCode;
                return;
            }
            // bitstreamToString(len * 8) + huffman_decode (:1261-1263, :468-482): every bit of
            // every byte, the zero padding included -- it may yield spurious trailing symbols.
            PlaneDecoder dec(t, payload[c], (unsigned long long)lens[c] * 8);
            n = dec.decode(outs[c], outs[c] ? caps[c] : 0, chunk_threads[c]);
        }
        *ns[c] = n; // the count found; more than cap means the caller's buffer was too small
        if (n > caps[c])
            status[c] = kTooSmall;
    };
    if ((size_t)lens[0] + (size_t)lens[1] + (size_t)lens[2] >= (1u << 18)) {
        parallel_chunks(3, [&](unsigned c) { decode_plane((int)c); });
    } else {
        for (int c = 0; c < 3; ++c)
            decode_plane(c);
    }
    for (int c = 0; c < 3; ++c)
        if (status[c] == kNoCode)
            return fail("bin_decode: a plane with fewer than two distinct symbols has no decodable code "
                        "(the reference dereferences a null child here, encode_decode.cpp:474)");
    for (int c = 0; c < 3; ++c)
        if (status[c] == kTooSmall)
            return fail("bin_decode: symbol buffer too small (the counts needed are in *n_y, *n_cb, *n_cr)");
    return 0;
}

int jpegamd_bin_encode(const int32_t *sym_y, size_t n_y, const int32_t *sym_cb, size_t n_cb,
                       const int32_t *sym_cr, size_t n_cr, int y_rows, int y_cols, int c_rows, int c_cols,
                       uint8_t *out, size_t cap, size_t *out_len)
{
    return jpegamd_guarded("bin_encode", [&] { return bin_encode_impl(sym_y, n_y, sym_cb, n_cb, sym_cr, n_cr, y_rows, y_cols, c_rows, c_cols, out, cap, out_len); });
}

int jpegamd_bin_decode(const uint8_t *bin, size_t len, int32_t dims[6],
                       int32_t *sym_y, size_t cap_y, size_t *n_y,
                       int32_t *sym_cb, size_t cap_cb, size_t *n_cb,
                       int32_t *sym_cr, size_t cap_cr, size_t *n_cr)
{
    return jpegamd_guarded("bin_decode", [&] { return bin_decode_impl(bin, len, dims, sym_y, cap_y, n_y, sym_cb, cap_cb, n_cb, sym_cr, cap_cr, n_cr); });
}

} // extern "C"
