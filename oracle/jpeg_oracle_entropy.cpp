/*
 * jpeg_oracle_entropy.cpp -- CPU restatement of the reference's entropy coder
 * and .bin container.  TEST INFRASTRUCTURE ONLY (see jpeg_oracle.h).
 *
 * The byte image of a .bin depends on libstdc++ behaviour the reference
 * inherits implicitly: iteration order of std::unordered_map<int,int> after a
 * given insertion history, and how std::priority_queue orders equal
 * frequencies.  This file therefore uses the same containers with the same
 * insertion sequence instead of re-deriving those orders.
 * Citations are relative to /root/reference.
 */
#include "jpeg_oracle.h"

#include <algorithm>
#include <cstring>
#include <queue>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Node { /* encode_decode.cpp:385-393 */
    int value;
    int frequency;
    Node *left;
    Node *right;
    Node(int v, int f) : value(v), frequency(f), left(nullptr), right(nullptr) {}
};

struct Greater { /* :395-401 */
    bool operator()(Node *a, Node *b) const { return a->frequency > b->frequency; }
};

struct Tree {
    std::vector<Node *> pool;
    Node *root = nullptr;
    ~Tree()
    {
        for (Node *n : pool)
            delete n;
    }
    Node *make(int v, int f)
    {
        pool.push_back(new Node(v, f));
        return pool.back();
    }
};

/* build_huffman_tree, encode_decode.cpp:403-442 */
void build_tree(const std::unordered_map<int, int> &freq, Tree &t)
{
    std::vector<std::pair<int, int>> fv(freq.begin(), freq.end());
    std::sort(fv.begin(), fv.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) {
        if (a.second == b.second)
            return a.first < b.first;
        return a.second < b.second;
    });
    std::priority_queue<Node *, std::vector<Node *>, Greater> heap;
    for (const auto &p : fv)
        heap.push(t.make(p.first, p.second));
    while (heap.size() > 1) {
        Node *l = heap.top();
        heap.pop();
        Node *r = heap.top();
        heap.pop();
        Node *in = t.make(99999, l->frequency + r->frequency);
        in->left = l;
        in->right = r;
        heap.push(in);
    }
    t.root = heap.empty() ? nullptr : heap.top();
}

/* generate_huffman_codes, :444-456 */
void gen_codes(Node *n, const std::string &code, std::unordered_map<int, std::string> &codes)
{
    if (!n)
        return;
    if (n->value != 99999)
        codes[n->value] = code;
    gen_codes(n->left, code + "0", codes);
    gen_codes(n->right, code + "1", codes);
}

struct Channel {
    std::unordered_map<int, int> freq;
    std::vector<unsigned char> bytes;
};

/* encodeChannel tail :648-658, then stringToBitstream :1093-1127 (the '0'/'1'
 * string is skipped: bits are packed MSB-first as they are produced). */
void encode_channel(const int32_t *sym, size_t n, Channel &ch)
{
    for (size_t i = 0; i < n; ++i) /* build_frequency_dict :375-383 */
        ch.freq[sym[i]]++;
    /* the reference copies the map twice before writing it (main.cpp:117,
     * encode_decode.cpp:658): a libstdc++ copy preserves iteration order. */
    Tree t;
    build_tree(ch.freq, t);
    std::unordered_map<int, std::string> codes;
    gen_codes(t.root, "", codes);
    unsigned char cur = 0;
    int pos = 0;
    for (size_t i = 0; i < n; ++i) {
        const std::string &c = codes.at(sym[i]);
        for (char bit : c) {
            if (bit == '1')
                cur |= (unsigned char)(1 << (7 - pos));
            if (++pos == 8) {
                ch.bytes.push_back(cur);
                cur = 0;
                pos = 0;
            }
        }
    }
    if (pos > 0)
        ch.bytes.push_back(cur);
}

void put_i32(std::vector<uint8_t> &o, int v)
{
    uint8_t b[4];
    std::memcpy(b, &v, 4); /* native little-endian, as file.write(&int) :1162 */
    o.insert(o.end(), b, b + 4);
}

} // namespace

extern "C" size_t orc_write_bin(const int32_t *sym_y, size_t n_y, const int32_t *sym_cb, size_t n_cb,
                                const int32_t *sym_cr, size_t n_cr,
                                int y_rows, int y_cols, int c_rows, int c_cols,
                                uint8_t *out, size_t cap)
{
    Channel y, cb, cr;
    encode_channel(sym_y, n_y, y);
    encode_channel(sym_cb, n_cb, cb);
    encode_channel(sym_cr, n_cr, cr);
    std::vector<uint8_t> o;
    /* saveEncodedData :1161-1217 */
    put_i32(o, y_rows);
    put_i32(o, y_cols);
    put_i32(o, c_rows);
    put_i32(o, c_cols);
    put_i32(o, c_rows);
    put_i32(o, c_cols);
    put_i32(o, (int)y.bytes.size());
    put_i32(o, (int)cb.bytes.size());
    put_i32(o, (int)cr.bytes.size());
    o.insert(o.end(), y.bytes.begin(), y.bytes.end());
    o.insert(o.end(), cb.bytes.begin(), cb.bytes.end());
    o.insert(o.end(), cr.bytes.begin(), cr.bytes.end());
    put_i32(o, (int)y.freq.size());
    put_i32(o, (int)cb.freq.size());
    put_i32(o, (int)cr.freq.size());
    for (const Channel *c : {&y, &cb, &cr})
        for (const auto &p : c->freq) {
            put_i32(o, p.first);
            put_i32(o, p.second);
        }
    if (o.size() > cap)
        return 0;
    std::memcpy(out, o.data(), o.size());
    return o.size();
}

namespace {
int get_i32(const uint8_t *p)
{
    int v;
    std::memcpy(&v, p, 4);
    return v;
}
} // namespace

extern "C" int orc_read_bin(const uint8_t *bin, size_t len, int32_t dims[6],
                            int32_t *sym_y, size_t cap_y, size_t *n_y,
                            int32_t *sym_cb, size_t cap_cb, size_t *n_cb,
                            int32_t *sym_cr, size_t cap_cr, size_t *n_cr)
{
    /* loadEncodedData :1236-1296 */
    if (len < 36)
        return 1;
    for (int i = 0; i < 6; ++i)
        dims[i] = get_i32(bin + 4 * i);
    int lens[3] = {get_i32(bin + 24), get_i32(bin + 28), get_i32(bin + 32)};
    size_t off = 36;
    const uint8_t *payload[3];
    for (int c = 0; c < 3; ++c) {
        if (lens[c] < 0 || off + (size_t)lens[c] > len)
            return 2;
        payload[c] = bin + off;
        off += (size_t)lens[c];
    }
    if (off + 12 > len)
        return 3;
    int dsz[3] = {get_i32(bin + off), get_i32(bin + off + 4), get_i32(bin + off + 8)};
    off += 12;
    int32_t *outs[3] = {sym_y, sym_cb, sym_cr};
    size_t caps[3] = {cap_y, cap_cb, cap_cr};
    size_t *ns[3] = {n_y, n_cb, n_cr};
    for (int c = 0; c < 3; ++c) {
        std::unordered_map<int, int> freq;
        for (int i = 0; i < dsz[c]; ++i) {
            if (off + 8 > len)
                return 4;
            int k = get_i32(bin + off), v = get_i32(bin + off + 4);
            off += 8;
            freq[k] = v; /* :1279 */
        }
        Tree t;
        build_tree(freq, t); /* decodeChannel :664 */
        if (!t.root || (!t.root->left && !t.root->right))
            return 5; /* single-symbol alphabet: the reference would crash (:474) */
        /* bitstreamToString(len*8) :1261 + huffman_decode :468-482: every bit
         * of every byte, pad bits included. */
        size_t n = 0;
        Node *cur = t.root;
        for (int b = 0; b < lens[c]; ++b)
            for (int i = 7; i >= 0; --i) {
                cur = (payload[c][b] & (1 << i)) ? cur->right : cur->left;
                if (!cur->left && !cur->right) {
                    if (n >= caps[c])
                        return 6;
                    outs[c][n++] = cur->value;
                    cur = t.root;
                }
            }
        *ns[c] = n;
    }
    return 0;
}
