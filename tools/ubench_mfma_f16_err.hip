// How exactly does v_mfma_f32_32x32x16_f16 accumulate?  The split-float16 shortlist of csrc/knn.hip charges every addition two
// units in the last place of the running sum of magnitudes (the ISA does not specify the order or the rounding of the adds inside
// the instruction).  This measures it: 32 x 32 dot products of K float16 pairs with magnitudes over several decades, chained over
// K / 16 instructions, against the exact sum (products of two 11-bit numbers are exact in double).  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

__global__ void k(const _Float16* a, const _Float16* b, int K, float* out)
{
    // A: 32 rows x K, B: 32 rows x K (row = lane % 32, the lane half takes k = 16 s + 8 h .. + 8): out[i][j] = sum_k A[i][k] B[j][k]
    const int lane = threadIdx.x, lr = lane & 31, lh = lane >> 5;
    f16v acc = {0};
    for (int s = 0; s < K / 16; s++) {
        const h8 av = *reinterpret_cast<const h8*>(a + lr * K + 16 * s + 8 * lh);
        const h8 bv = *reinterpret_cast<const h8*>(b + lr * K + 16 * s + 8 * lh);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    }
    for (int r = 0; r < 16; r++) out[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + lr] = acc[r];
}

int main()
{
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd(0, 1);
    std::uniform_real_distribution<double> ud(-6, 3);
    double worst = 0, worst_den = 0;
    for (int K : {16, 208, 624, 2048}) {
        for (int trial = 0; trial < 200; trial++) {
            std::vector<_Float16> a(32 * K), b(32 * K);
            const bool denorm = trial % 4 == 3;       // a quarter of the trials with float16 DENORMAL operands mixed in
            for (auto& v : a) v = (_Float16)(nd(rng) * std::pow(10.0, ud(rng)) * (denorm && (rng() & 3) == 0 ? 1e-6 : 1.0));
            for (auto& v : b) v = (_Float16)(nd(rng) * std::pow(10.0, ud(rng)));
            _Float16 *da, *db; float* dout;
            hipMalloc(&da, a.size() * 2); hipMalloc(&db, b.size() * 2); hipMalloc(&dout, 32 * 32 * 4);
            hipMemcpy(da, a.data(), a.size() * 2, hipMemcpyHostToDevice);
            hipMemcpy(db, b.data(), b.size() * 2, hipMemcpyHostToDevice);
            k<<<1, 64>>>(da, db, K, dout);
            std::vector<float> out(32 * 32);
            hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
            for (int i = 0; i < 32; i++)
                for (int j = 0; j < 32; j++) {
                    double s = 0, m = 0;
                    for (int kk = 0; kk < K; kk++) { const double p = (double)a[i * K + kk] * (double)b[j * K + kk]; s += p; m += std::fabs(p); }
                    const double e = std::fabs((double)out[i * 32 + j] - s) / (m > 0 ? m : 1) / K / 5.9604644775390625e-8;
                    if (denorm) { if (e > worst_den) worst_den = e; } else if (e > worst) worst = e;
                }
            hipFree(da); hipFree(db); hipFree(dout);
        }
        printf("K = %4d: worst |mfma - exact| / (sum |terms| * K * 2^-24) so far: %.4f (normal operands)  %.4f (with float16 denormals)\n",
               K, worst, worst_den);
    }
    printf("the shortlist's bound charges 2.0 per addition\n");
    return 0;
}
