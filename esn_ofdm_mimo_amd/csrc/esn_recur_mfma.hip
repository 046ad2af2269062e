// Geometry table and precision dispatch of the MFMA recurrence kernels; the kernels themselves are
// instantiated per precision in esn_recur_mfma_{f32,f16,bf16}.hip (compiled in parallel).
#include <stdio.h>
#include <stdlib.h>
#include "esn_common.h"

namespace esn {

int launch_recur_mfma_f32(const RecurParams& p, hipStream_t stream);
int launch_recur_mfma_f16(const RecurParams& p, hipStream_t stream);
int launch_recur_mfma_bf16(const RecurParams& p, hipStream_t stream);

// Tiling table: (NW, MT, NT) per precision and reservoir size.  Constraints:
// Mp = 32*MT*NW >= n_res; LDS = 32*NT*Ks*ES <= 160 KiB; accumulators 16*MT*NT
// VGPRs/lane within the 512/(waves per SIMD) budget; NT*2 <= NW (column owners).
bool mfma_geometry(int precision, int n_res, int n_in, int n_out, bool harvest, Geometry* g) {
    const int es = (precision == ESN_F32) ? 4 : 2;
    int NW, MT, NT;
    if (precision == ESN_F32) {
        if (n_res <= 128)       { NW = 4;  MT = 1; NT = 2; }
        else if (n_res <= 256)  { NW = 4;  MT = 2; NT = 2; }
        else if (n_res <= 512)  { NW = 8;  MT = 2; NT = 2; }
        else if (n_res <= 1024) { NW = 8;  MT = 4; NT = 1; }
        else return false;
    } else {
        if (n_res <= 128)       { NW = 4;  MT = 1; NT = 2; }
        else if (n_res <= 256)  { NW = 8;  MT = 1; NT = 4; }
        else if (n_res <= 512)  { NW = 8;  MT = 2; NT = 4; }
        else if (n_res <= 1024) { NW = 8;  MT = 4; NT = 2; }
        else if (n_res <= 2048) { NW = 16; MT = 4; NT = 1; }
        else return false;
    }
    // The packed weight image is a function of (Mp, Kp) only, and Mp comes from the table above: pack,
    // harvest and predict always agree on it.  A tuning override ("NW,MT,NT", benchmarks only; knobs())
    // may re-cut the same Mp rows into other waves / tiles but is IGNORED when it would change Mp.
    const int Mp_table = 32 * MT * NW;
    if (!harvest) {
        const int* ov = (es == 2) ? knobs().geom16 : knobs().geom32;
        if (ov[0] > 0 && 32 * ov[0] * ov[1] == Mp_table) { NW = ov[0]; MT = ov[1]; NT = ov[2]; }
    }
    // harvest: one pilot per trained ESN -> few sequences; a 32-frame tile spreads them over more CUs
    if (harvest && n_res > 256 && n_res <= 512) { NW = 8; MT = 2; NT = 1; }
    if (n_out > 16) return false;
    g->NW = NW; g->MT = MT; g->NT = NT;
    g->Mp = 32 * MT * NW;
    g->kin = g->Mp;
    g->kfb = g->kin + round_up(n_in, 4);
    // (reservoirs beyond 1024: K padded to whole 64-deep chunks of the launch-per-step GEMM, esn_recur_big.hip)
    g->Kp = round_up(g->kfb + round_up(n_out, 4), (es == 2 && n_res > 1024) ? 64 : 32);
#ifdef ESN_WITH_RS
    g->rs = (es == 2 && g->Mp == 512 && g->Kp == 544 && g->kfb - g->kin == 16 && n_out <= 8) ? 1 : 0;
#else
    g->rs = 0;      // the register-state kernel (and its read-out image) exist only in ESN_WITH_RS=1 builds
#endif
    // the 16x16x32 skewed predict kernel (esn_recur_skew16_impl.h): a function of the shape alone -- pack, harvest and
    // predict agree on the images whatever the knobs say
    g->s16 = (es == 2 && g->Mp == 512 && g->Kp == 544 && n_out <= 8 && !g->rs &&
              (n_in == 2 || n_in == 4 || n_in == 8 || n_in == 16)) ? 1 : 0;
    g->big = (es == 2 && n_res > 1024 && n_in <= 16 && n_out <= 8 && g->Mp % 256 == 0 && g->Mp <= 2048 &&
              g->kfb - g->kin + round_up(n_out, 4) <= 32) ? 1 : 0;
    // row stride: an odd number of 16-byte slots -> conflict-free b128 column reads
    int slots = g->Kp * es / 16;
    g->Ks = g->Kp + ((slots % 2 == 0) ? 16 / es : 0);
    g->Bt = 32 * NT;
    g->ro_fold = (es == 2 && n_out <= 8) ? 1 : 0;
    g->ro_parts = (es == 2 && !g->ro_fold) ? 2 : 1;
    // skewed wave schedule (predict): 8-wave fp16/bf16 tilings with one readout image and state
    // k-groups that split into two even halves; ESN_SKEW=0 keeps the in-step schedule (A/B runs)
    {
        const int nkgS = g->Mp * es / 32;
        const int n_uf = g->Kp * es / 32 - nkgS;
        g->skew = (!harvest && es == 2 && NW == 8 && (NT == 2 || NT == 4) && g->ro_parts == 1 && nkgS % 8 == 0 && n_uf <= 4 &&
                   (n_in == 2 || n_in == 4 || n_in == 8 || n_in == 16) &&
                   knobs().skew) ? 1 : 0;
    }
    // Zt image + the small frame / scale tables behind it (esn_recur_mfma_impl.h); the skewed schedule
    // stages the raw input rows in one 1 KB slot per DMA instruction (two per 16-frame tile)
    auto lds_bytes = [&](int skew) {
        const size_t staging = skew ? (size_t)(g->Bt / 16) * 2048 : 8 * (size_t)g->Bt * n_in;
        return (size_t)g->Bt * g->Ks * es + 4 * (size_t)g->Bt + 8 * (size_t)(g->Bt / 16) * ((g->kfb - g->kin) + 16)
               + staging + 4 * (size_t)g->Bt;     // + input offsets of the frames
    };
    if (g->skew && lds_bytes(1) > 160 * 1024) g->skew = 0;
    return lds_bytes(g->skew) <= 160 * 1024;
}

int launch_recur_mfma(int precision, const RecurParams& p, hipStream_t stream) {
    if (precision == ESN_F32) return launch_recur_mfma_f32(p, stream);
    if (precision == ESN_F16) return launch_recur_mfma_f16(p, stream);
    if (precision == ESN_BF16) return launch_recur_mfma_bf16(p, stream);
    return -1;
}

}  // namespace esn
