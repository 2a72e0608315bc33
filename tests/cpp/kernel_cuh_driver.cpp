// Test driver for the kernel.cuh drop-in symbols: reads blocks + table from a
// file, calls the two C++ entry points exactly as encode_decode.cpp:728 / :790
// do, writes the results.  Usage: kernel_cuh_driver <in.bin> <out.bin>
//   in : int32 nb | int32 q[64] | int32 pixels[nb*64] | float coef[nb*64]
//   out: float quantised[nb*64] | int32 restored[nb*64]
#include <cstdint>
#include <cstdio>
#include <vector>
using namespace std;
void dctQuantizationParallel(const vector<vector<vector<int>>> &, const vector<vector<int>> &, vector<vector<float>> &);
void dequantizeInverseDCTParallel(const vector<vector<vector<float>>> &, const vector<vector<int>> &,
                                  vector<vector<vector<int>>> &);
int main(int argc, char **argv)
{
    if (argc != 3)
        return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f)
        return 3;
    int32_t nb = 0, q[64];
    if (fread(&nb, 4, 1, f) != 1 || fread(q, 4, 64, f) != 64)
        return 4;
    vector<int32_t> px((size_t)nb * 64);
    vector<float> cf((size_t)nb * 64);
    if (fread(px.data(), 4, px.size(), f) != px.size() || fread(cf.data(), 4, cf.size(), f) != cf.size())
        return 4;
    fclose(f);
    vector<vector<int>> table(8, vector<int>(8));
    for (int i = 0; i < 64; ++i)
        table[i / 8][i % 8] = q[i];
    vector<vector<vector<int>>> blocks(nb, vector<vector<int>>(8, vector<int>(8)));
    vector<vector<vector<float>>> coefs(nb, vector<vector<float>>(8, vector<float>(8)));
    for (int b = 0; b < nb; ++b)
        for (int i = 0; i < 64; ++i) {
            blocks[b][i / 8][i % 8] = px[(size_t)b * 64 + i];
            coefs[b][i / 8][i % 8] = cf[(size_t)b * 64 + i];
        }
    vector<vector<float>> quantised;
    vector<vector<vector<int>>> restored;
    dctQuantizationParallel(blocks, table, quantised);
    dequantizeInverseDCTParallel(coefs, table, restored);
    if ((int)quantised.size() != nb || (int)restored.size() != nb)
        return 5;
    f = fopen(argv[2], "wb");
    for (int b = 0; b < nb; ++b)
        fwrite(quantised[b].data(), 4, 64, f);
    for (int b = 0; b < nb; ++b)
        for (int i = 0; i < 8; ++i)
            fwrite(restored[b][i].data(), 4, 8, f);
    fclose(f);
    return 0;
}
