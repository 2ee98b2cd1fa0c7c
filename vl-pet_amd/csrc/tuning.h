// Experiment switches of the kernels (alternative kernels / plans kept for A/B measurements).
// The product library reads NOTHING from the environment: every switch below is a constant.  Only a diagnosis build
// (make DEBUG=1: -DVLPET_DEBUG, what tools/ uses for same-box A/Bs and what the tests of the alternatives ask for via
// vlpet_debug_build()) fills the table from VLPET_* variables, once, when the library is loaded.
#pragma once
#include <cstdlib>

struct VlpetTuning {
    int rg = 0;             // VLPET_RG: force the row groups per workgroup of the row kernels (0: pick_row_groups)
    int bwd2 = 1;           // VLPET_BWD2=0: single-wave row kernel instead of the chain-split one
    int bwd3 = -1;          // VLPET_BWD3=1|0: older two-pass form always / never (-1: r = 192 and M <= 4096)
    int bwd3_units = 0;     // VLPET_BWD3_UNITS: row chunks of the older pass 2
    int bwd3_form = 1;      // VLPET_BWD3_FORM: 0 = four-role form of the older pass 2
    int k4_waves4 = 0;      // VLPET_K4_WAVES4=1: the 4-wave K4 forward
    int wgrad_wgs = 0;      // VLPET_WGRAD_WGS: workgroup target of the weight-gradient plan (0: one per CU, XCD-bounded)
    int wgrad_tr = 1;       // VLPET_WGRAD_TR=0: identity-transpose weight-gradient kernel
    int wgrad_stream = 1;   // VLPET_WGRAD_STREAM=0: per-wave transpose-read kernel instead of the streaming one
    int wgrad_nstg = 3;     // VLPET_WGRAD_NSTG=4: ring depth of the streaming kernel
    int wgs_mode = 0;       // VLPET_WGS_MODE: ablation bits of the streaming kernel (needs -DVLPET_WGRAD_EXP as well)
    int attn_occ = 0;       // VLPET_ATTN_OCC=2|3: waves per SIMD of the attention backward (0: by shape)
    int attn_nw = 0;        // VLPET_ATTN_NW=6: six-wave attention backward
    int dz2 = 1;            // VLPET_DZ2=0: chain-split pass 1 of the K1 backward (pet_gate_dz_kernel) instead of the feature-split one
    int dz2_fsplit = 0;     // VLPET_DZ2_FSPLIT=1|2|4: feature blocks of pass 1 (0: by shape)
    int dz6 = 1;            // VLPET_DZ6=0: pet_gate_dz_kernel instead of the four-wave feature-split pass 1 at six tiles; 2: that pass at every size
    int k4_wgrad2 = 1;      // VLPET_K4_WGRAD2=0: K4 weight gradient as jobs of the generic stream instead of the tiled split-K GEMM
    int ng2 = 1;            // VLPET_NG2=0: row kernel + streaming weight gradients for the backward without a gate instead of the two-pass form
    int fwd2p = -1;         // VLPET_FWD2P: gated forward of the training form: -1 by shape (k1_fwd2p_preferred), 0 one-kernel (pet_gate_fwd.hip), 1 two-pass
                            //   (pet_fwd2p.hip), 2 = pass A only, 3 = pass B only (timing)
    int cols_red = 1;       // VLPET_COLS_RED=0: pass 2 of the gated K1 backward always leaves partial slabs for a finalize launch (round 3) instead of
                            //   summing its row chunks inside the launch (round 6, cols_reduce.h: from 8,192 rows); 2: inside the launch at every size
    int dbg = 0;            // VLPET_DBG: ablation / stamp bits
};

#ifdef VLPET_DEBUG
inline const VlpetTuning& vlpet_tuning() {
    static const VlpetTuning t = [] {
        VlpetTuning v;
        auto rd = [](const char* n, int& dst) { if (const char* e = getenv(n)) dst = atoi(e); };
        rd("VLPET_RG", v.rg); rd("VLPET_BWD2", v.bwd2); rd("VLPET_BWD3", v.bwd3); rd("VLPET_BWD3_UNITS", v.bwd3_units);
        rd("VLPET_BWD3_FORM", v.bwd3_form); rd("VLPET_K4_WAVES4", v.k4_waves4); rd("VLPET_WGRAD_WGS", v.wgrad_wgs);
        rd("VLPET_WGRAD_TR", v.wgrad_tr); rd("VLPET_WGRAD_STREAM", v.wgrad_stream); rd("VLPET_WGRAD_NSTG", v.wgrad_nstg);
        rd("VLPET_WGS_MODE", v.wgs_mode); rd("VLPET_ATTN_OCC", v.attn_occ); rd("VLPET_ATTN_NW", v.attn_nw); rd("VLPET_DBG", v.dbg); rd("VLPET_DZ2", v.dz2); rd("VLPET_DZ6", v.dz6); rd("VLPET_DZ2_FSPLIT", v.dz2_fsplit); rd("VLPET_K4_WGRAD2", v.k4_wgrad2); rd("VLPET_NG2", v.ng2); rd("VLPET_FWD2P", v.fwd2p); rd("VLPET_COLS_RED", v.cols_red);
        return v;
    }();
    return t;
}
#define VLPET_IS_DEBUG_BUILD 1
#else
inline const VlpetTuning& vlpet_tuning() { static const VlpetTuning t{}; return t; }
#define VLPET_IS_DEBUG_BUILD 0
#endif
