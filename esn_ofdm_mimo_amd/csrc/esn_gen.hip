// On-device Monte-Carlo frame generator: the producer directly upstream of the
// detector.  float64 / complex128 like the reference; random streams are counter based
// (Philox4x32-10 keyed by seed, counter = (frame or link index, purpose, position)), so a frame is
// the same on any rank and any launch shape.  Every random input can also be supplied by the
// caller (bits_in / noise_in / gains_in): that is the deterministic parity mode the tests use
// against oracle/ofdm_frames.py.
//
//   gen_taps_kernel    TDL-B impulse responses   Demo_MIMO_4x8_Sionna_CDL_ESN_v2.py:127-177
//                      exponential-PDP Rayleigh  OFDM_MIMO_2-2_NBF_LDPC.py:162-164,272-279
//   gen_frames_kernel  bits -> QAM (:406-411) -> N*ifft (:416) -> CP (:417) -> sqrt(Pi) -> PA (:419)
//                      -> per-link FIR, zero initial state (:422-425) -> AWGN (:426)
#include "esn_common.h"
#include <type_traits>

namespace esn {

struct Philox {
    uint32_t k0, k1;
    __device__ __forceinline__ void round(uint32_t (&c)[4], uint32_t ka, uint32_t kb) const {
        const uint64_t p0 = (uint64_t)0xD2511F53U * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57U * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ ka;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ kb;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
    }
    __device__ __forceinline__ void operator()(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                               uint32_t (&out)[4]) const {
        uint32_t c[4] = {c0, c1, c2, c3};
        uint32_t ka = k0, kb = k1;
#pragma unroll
        for (int i = 0; i < 10; ++i) { round(c, ka, kb); ka += 0x9E3779B9U; kb += 0xBB67AE85U; }
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = c[i];
    }
};

// two independent N(0,1) from two 32-bit words (Box-Muller, float64)
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, double& z0, double& z1) {
    const double u1 = ((double)a + 0.5) * (1.0 / 4294967296.0);
    const double u2 = ((double)b + 0.5) * (1.0 / 4294967296.0);
    const double rr = sqrt(-2.0 * log(u1));
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    z0 = rr * cs; z1 = rr * sn;
}

// the same from the float32 hardware transcendentals (v_log_f32, v_sqrt_f32, v_sin_f32 / v_cos_f32 take the angle
// in revolutions): ~10 instructions instead of ~150 of float64 library code per pair.  Relative accuracy ~1e-6 --
// a noise sample needs the right DISTRIBUTION, not float64 digits; |z| reaches 6.7 (u1 >= 2^-33).  The AWGN of
// 1 080 samples per frame was two thirds of the generator's instructions in float64.
__device__ __forceinline__ void box_muller_fast(uint32_t a, uint32_t b, double& z0, double& z1) {
    const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);          // 24 bits: exact in float32, never 0 or 1
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    const float rr = __builtin_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));   // sqrt(-2 ln u1), log2 based
    z0 = (double)(rr * __builtin_amdgcn_cosf(u2));
    z1 = (double)(rr * __builtin_amdgcn_sinf(u2));
}

enum { PURPOSE_BITS = 1, PURPOSE_NOISE = 2, PURPOSE_TAPS = 3 };

__global__ void gen_taps_kernel(TapParams tp) {
    const int link = blockIdx.x * blockDim.x + threadIdx.x;
    if (link >= tp.n_links) return;
    const Philox ph{(uint32_t)tp.seed, (uint32_t)(tp.seed >> 32)};
    const uint64_t gl = tp.link_offset + link;
    double hr[16], hi[16];
    for (int k = 0; k < tp.isi; ++k) { hr[k] = 0.0; hi[k] = 0.0; }
    const int np = (tp.kind == 2) ? 1 : tp.n_paths;
    for (int pth = 0; pth < np; ++pth) {
        double gr, gi;
        if (tp.gains_in) {
            gr = tp.gains_in[((size_t)link * tp.n_paths + pth) * 2];
            gi = tp.gains_in[((size_t)link * tp.n_paths + pth) * 2 + 1];
        } else {
            uint32_t w[4];
            ph((uint32_t)gl, (uint32_t)(gl >> 32), PURPOSE_TAPS, (uint32_t)pth, w);
            box_muller(w[0], w[1], gr, gi);
        }
        if (tp.kind == 0) {          // CN(0, p) path gain split between floor / ceil taps
            const double a = 0.70710678118654752 * tp.path_sqrt_pow[pth];
            gr *= a; gi *= a;
            const double d = tp.path_delay_samples[pth];           // delay in samples
            const int i0 = (int)floor(d);
            const double frac = d - (double)i0;
            if (i0 >= 0 && i0 < tp.isi) { hr[i0] = fma(gr, 1.0 - frac, hr[i0]); hi[i0] = fma(gi, 1.0 - frac, hi[i0]); }
            if (i0 + 1 >= 0 && i0 + 1 < tp.isi) { hr[i0 + 1] = fma(gr, frac, hr[i0 + 1]); hi[i0 + 1] = fma(gi, frac, hi[i0 + 1]); }
        } else if (tp.kind == 1) {   // one CN(0, pdp_k) tap per path
            hr[pth] = gr * 0.70710678118654752 * tp.path_sqrt_pow[pth];
            hi[pth] = gi * 0.70710678118654752 * tp.path_sqrt_pow[pth];
        } else {                     // flat channel, random phase, unit modulus
            const double a = sqrt(gr * gr + gi * gi);
            hr[0] = a > 0 ? gr / a : 1.0; hi[0] = a > 0 ? gi / a : 0.0;
        }
    }
    if (tp.kind == 0) {              // unit energy per link (driver :162-164)
        double e = 0.0;
        for (int k = 0; k < tp.isi; ++k) e += hr[k] * hr[k] + hi[k] * hi[k];
        if (e > 0.0) { const double s = 1.0 / sqrt(e); for (int k = 0; k < tp.isi; ++k) { hr[k] *= s; hi[k] *= s; } }
    }
    for (int k = 0; k < tp.isi; ++k) {
        tp.taps[((size_t)link * tp.isi + k) * 2] = hr[k];
        tp.taps[((size_t)link * tp.isi + k) * 2 + 1] = hi[k];
    }
}

// One workgroup per frame.  LDS: X[n_t][N] (frequency -> time in place), xpa[n_t][T] (post-PA signal),
// tw[N/2] (twiddles, one table per workgroup instead of a sincospi per butterfly).  The channel is the
// hot part -- T n_r outputs of n_t isi complex MACs each -- and is register-blocked: a wave takes 64
// consecutive time samples and four receive antennas, so every x[t-k][tx] is read from LDS once per FOUR
// outputs (consecutive lanes, consecutive addresses: conflict-free) and the taps are wave-uniform, i.e.
// scalar loads straight from the tap array (round 2 kept them in LDS and read them with an 8-way bank
// conflict per MAC: 75 % of the LDS cycles were conflicts, 6.2 ms per 153 600 frames).
constexpr int GEN_RXG = 4;      // receive antennas per wave task
typedef uint32_t u32x4v __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void gen_frames_kernel(FrameGenParams fp) {
    extern __shared__ __attribute__((aligned(16))) char gsm[];
    const int N = fp.n_sub, T = N + fp.cp, n_t = fp.n_t, n_r = fp.n_r, isi = fp.isi, m = fp.m;
    double2* X = reinterpret_cast<double2*>(gsm);            // [n_t][N]   freq -> time (in place)
    double2* xpa = X + (size_t)n_t * N;                       // [n_t][T]   post-PA time signal
    double2* tw = xpa + (size_t)n_t * T;                      // [N/2]      exp(+2 pi i k / N)
    double2* ctap = tw + (N >> 1);                            // [n_r][n_t][isi] this block's taps
    const int tid = threadIdx.x, nth = blockDim.x;
    const int frame = blockIdx.x;
    const int blk = frame / fp.frames_per_block;
    const uint64_t gf = fp.frame_offset + frame;
    const Philox ph{(uint32_t)fp.seed, (uint32_t)(fp.seed >> 32)};
    const int side = 1 << (m / 2);
    const double norm = sqrt(2.0 * (double)(side * side - 1) / 3.0);
    const int half = N >> 1;

    for (int k = tid; k < half; k += nth) {
        double sn, cs;
        sincospi(2.0 * (double)k / (double)N, &sn, &cs);
        tw[k] = make_double2(cs, sn);
    }
    for (int i = tid; i < n_r * n_t * isi; i += nth)
        ctap[i] = reinterpret_cast<const double2*>(fp.taps)[(size_t)blk * n_r * n_t * isi + i];
    // ---- bits -> constellation point (index = sum_b bit_b 2^b = i*side + j; Re = pam[i], Im = pam[j])
    // random payload: ONE Philox call per 128 bits = 4 (32 / m) symbols (v_mul_hi_u32 is a quarter-rate instruction:
    // a call per symbol was 512 calls per frame for 2 048 bits); thread t owns call t
    const int spw = 32 / m, spc = 4 * spw;                   // symbols per word / per call
    auto place = [&](int e, uint32_t idx) {
        const int sc = e / n_t, tx = e % n_t;
        for (int b = 0; b < m; ++b)
            fp.bits[((size_t)frame * N * m + (size_t)sc * m + b) * n_t + tx] = (uint8_t)((idx >> b) & 1);
        const int pi_ = (int)(idx / side), pj = (int)(idx % side);
        const int rv = (int)(__brev((unsigned)sc) >> (32 - fp.log2n));
        const bool on = !fp.ls_pattern || (sc % n_t == tx);
        X[(size_t)tx * N + rv] = on ? make_double2((2.0 * pi_ - (side - 1)) / norm, (2.0 * pj - (side - 1)) / norm)
                                    : make_double2(0.0, 0.0);
    };
    if (fp.bits_in) {
        for (int e = tid; e < N * n_t; e += nth) {
            const int sc = e / n_t, tx = e % n_t;
            uint32_t idx = 0;
            for (int b = 0; b < m; ++b)
                idx |= (uint32_t)(fp.bits_in[((size_t)frame * N * m + (size_t)sc * m + b) * n_t + tx] & 1) << b;
            place(e, idx);
        }
    } else if (m == 4 && n_t == 4) {
        // headline shape: a call's 32 symbols are 8 whole subcarriers; a subcarrier's 16 bit-bytes (bit b of tx at
        // b n_t + tx) leave as ONE 16-byte store instead of 16 byte stores (2 048 byte stores per frame were half of
        // the kernel's floor)
        for (int call = tid; call * 32 < N * 4; call += nth) {
            uint32_t w[4];
            ph((uint32_t)gf, (uint32_t)(gf >> 32), PURPOSE_BITS, (uint32_t)call, w);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const int sc = call * 8 + q * 2 + h2;
                    const uint32_t four = (w[q] >> (16 * h2)) & 0xffffu;          // the subcarrier's four 4-bit symbols
                    u32x4v ob;
#pragma unroll
                    for (int b = 0; b < 4; ++b)                                   // word b: bit b of tx 0..3, one byte each
                        ob[b] = ((four >> b) & 1u) | (((four >> (4 + b)) & 1u) << 8) | (((four >> (8 + b)) & 1u) << 16) |
                                (((four >> (12 + b)) & 1u) << 24);
                    *reinterpret_cast<u32x4v*>(fp.bits + ((size_t)frame * N + sc) * 16) = ob;
                    const int rv = (int)(__brev((unsigned)sc) >> (32 - fp.log2n));
#pragma unroll
                    for (int tx = 0; tx < 4; ++tx) {
                        const uint32_t idx = (four >> (4 * tx)) & 15u;
                        const bool on = !fp.ls_pattern || ((sc & 3) == tx);
                        X[(size_t)tx * N + rv] = on ? make_double2((2.0 * (int)(idx >> 2) - 3.0) / norm, (2.0 * (int)(idx & 3) - 3.0) / norm)
                                                    : make_double2(0.0, 0.0);
                    }
                }
        }
    } else {
        for (int call = tid; call * spc < N * n_t; call += nth) {
            uint32_t w[4];
            ph((uint32_t)gf, (uint32_t)(gf >> 32), PURPOSE_BITS, (uint32_t)call, w);
#pragma unroll
            for (int q = 0; q < 4; ++q)
                for (int j = 0; j < spw; ++j) {
                    const int e = call * spc + q * spw + j;
                    if (e < N * n_t) place(e, (w[q] >> (j * m)) & ((1u << m) - 1));
                }
        }
    }
    __syncthreads();
    // ---- x = N * ifft(X): un-normalised inverse DFT, radix-2 DIT on bit-reversed input, per tx
    for (int s = 1; s <= ((fp.ko & 4) ? 0 : fp.log2n); ++s) {
        const int hm = 1 << (s - 1), tstep = N >> s;
        for (int e = tid; e < n_t * half; e += nth) {
            const int tx = e / half, b = e % half;
            const int j = b & (hm - 1);
            const int base = ((b >> (s - 1)) << s) + j;
            const double2 w = tw[j * tstep];                 // exp(+2 pi i j / 2^s), same argument as sincospi(j / hm)
            double2* xx = X + (size_t)tx * N;
            const double2 a = xx[base], c = xx[base + hm];
            const double tr = c.x * w.x - c.y * w.y, ti = c.x * w.y + c.y * w.x;
            xx[base] = make_double2(a.x + tr, a.y + ti);
            xx[base + hm] = make_double2(a.x - tr, a.y - ti);
        }
        __syncthreads();
    }
    // ---- cyclic prefix, power scaling, PA
    const double sp = sqrt(fp.p_i[blk]), aclip = fp.a_clip[blk];
    for (int e = tid; e < T * n_t; e += nth) {
        const int t = e / n_t, tx = e % n_t;
        const int n = (t < fp.cp) ? (N - fp.cp + t) : (t - fp.cp);
        const double2 v = X[(size_t)tx * N + n];
        const double xr = v.x * sp, xi = v.y * sp;
        if (fp.x_cp) {
            fp.x_cp[((size_t)frame * T + t) * n_t * 2 + 2 * tx] = xr;
            fp.x_cp[((size_t)frame * T + t) * n_t * 2 + 2 * tx + 1] = xi;
        }
        const double mag = (fp.ko & 8) ? 0.0 : sqrt(xr * xr + xi * xi) / aclip;
        const double gpa = (fp.ko & 8) ? 1.0 : 1.0 / sqrt(1.0 + mag * mag);
        xpa[(size_t)tx * T + t] = make_double2(xr * gpa, xi * gpa);
    }
    __syncthreads();
    // ---- y[t][rx] = sum_tx sum_k c[rx][tx][k] x_pa[t-k][tx] + sqrt(T No / 2) (n_re + j n_im)
    const double sig = sqrt((double)T * fp.no * 0.5);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nth >> 6;
    const int t_chunks = (T + 63) >> 6, rx_groups = (n_r + GEN_RXG - 1) / GEN_RXG;
    for (int task = wave; task < t_chunks * rx_groups; task += nwaves) {
        const int rx0 = __builtin_amdgcn_readfirstlane((task / t_chunks) * GEN_RXG);
        const int t = (task % t_chunks) * 64 + lane;
        const bool live = t < T;
        double yr[GEN_RXG], yi[GEN_RXG];
#pragma unroll
        for (int q = 0; q < GEN_RXG; ++q) { yr[q] = 0.0; yi[q] = 0.0; }
        // taps: every lane of the wave reads the SAME LDS address (a broadcast: one LDS pass per lane group, no bank
        // conflict -- round 2 read them per (rx, tx, k) with rx varying across lanes: 8-way conflicts).  (Scalar loads
        // through the constant address space were tried first: correct, 16 dependent s_load round trips per task.)
        auto channel = [&](auto isi_tag) {
            constexpr int ISI = decltype(isi_tag)::value;            // 0 = run-time tap count
            const int ntap = ISI ? ISI : isi;
            for (int tx = 0; tx < n_t; ++tx) {
                const double2* xs = xpa + (size_t)tx * T;
                if constexpr (ISI > 0) {
                    double2 xv[ISI];
#pragma unroll
                    for (int k = 0; k < ISI; ++k)                      // zero initial state (lfilter)
                        xv[k] = (live && k <= t) ? xs[t - k] : make_double2(0.0, 0.0);
#pragma unroll
                    for (int q = 0; q < GEN_RXG; ++q) {
                        const int rx = rx0 + q < n_r ? rx0 + q : n_r - 1;     // (uniform; clamped, result unused)
                        const double2* c = ctap + ((size_t)rx * n_t + tx) * ISI;
#pragma unroll
                        for (int k = 0; k < ISI; ++k) {
                            const double2 cv = c[k];
                            // (explicit FMAs: -ffp-contract=off would make a complex MAC 4 mul + 4 add; the float64 FMA rate
                            //  bounds this loop)
                            yr[q] = fma(-cv.y, xv[k].y, fma(cv.x, xv[k].x, yr[q]));
                            yi[q] = fma(cv.y, xv[k].x, fma(cv.x, xv[k].y, yi[q]));
                        }
                    }
                } else {
                    for (int k = 0; k < ntap; ++k) {
                        const double2 xv = (live && k <= t) ? xs[t - k] : make_double2(0.0, 0.0);
#pragma unroll
                        for (int q = 0; q < GEN_RXG; ++q) {
                            const int rx = rx0 + q < n_r ? rx0 + q : n_r - 1;
                            const double2 cv = ctap[((size_t)rx * n_t + tx) * isi + k];
                            yr[q] = fma(-cv.y, xv.y, fma(cv.x, xv.x, yr[q]));
                            yi[q] = fma(cv.y, xv.x, fma(cv.x, xv.y, yi[q]));
                        }
                    }
                }
            }
        };
        if (!(fp.ko & 2)) { if (isi == 8) channel(std::integral_constant<int, 8>{}); else channel(std::integral_constant<int, 0>{}); }
        if (!live) continue;
        // AWGN: one Philox call per PAIR of receive antennas (4 words = two complex samples), counter (t, rx / 2)
#pragma unroll
        for (int q = 0; q < GEN_RXG; q += 2) {
            const int rx = rx0 + q;
            if (rx >= n_r) break;
            double nr[2], ni[2];
            if (fp.ko & 1) { nr[0] = nr[1] = ni[0] = ni[1] = 0.0; }
            else if (fp.noise_in) {
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    const int rxd = rx + d < n_r ? rx + d : rx;
                    nr[d] = fp.noise_in[((size_t)frame * T + t) * n_r * 2 + 2 * rxd];
                    ni[d] = fp.noise_in[((size_t)frame * T + t) * n_r * 2 + 2 * rxd + 1];
                }
            } else {
                uint32_t w[4];
                ph((uint32_t)gf, (uint32_t)(gf >> 32), PURPOSE_NOISE, (uint32_t)(t * ((n_r + 1) >> 1) + (rx >> 1)), w);
                box_muller_fast(w[0], w[1], nr[0], ni[0]);
                box_muller_fast(w[2], w[3], nr[1], ni[1]);
            }
#pragma unroll
            for (int d = 0; d < 2; ++d)
                if (rx + d < n_r)
                    reinterpret_cast<double2*>(fp.y_cp)[((size_t)frame * T + t) * n_r + rx + d] =
                        make_double2(yr[q + d] + sig * nr[d], yi[q + d] + sig * ni[d]);
        }
    }
}

int launch_gen_taps(const TapParams& tp, hipStream_t stream) {
    const int threads = 128;
    hipLaunchKernelGGL(gen_taps_kernel, dim3((tp.n_links + threads - 1) / threads), dim3(threads), 0, stream, tp);
    return (int)hipGetLastError();
}

int launch_gen_frames(const FrameGenParams& fp, hipStream_t stream) {
    const int T = fp.n_sub + fp.cp;
    const size_t lds = sizeof(double2) * ((size_t)fp.n_t * fp.n_sub + (size_t)fp.n_t * T + fp.n_sub / 2 +
                                          (size_t)fp.n_r * fp.n_t * fp.isi);
    if (lds > 150 * 1024) return -1;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gen_frames_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(gen_frames_kernel, dim3(fp.n_frames), dim3(256), lds, stream, fp);
    return (int)hipGetLastError();
}

}  // namespace esn
