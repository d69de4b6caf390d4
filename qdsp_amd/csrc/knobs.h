// knobs.h -- the library's tuning / experiment switches (QDSP_HIP_* environment variables), read ONCE.
//
// Round 3: `process_dev` used to call getenv + atoi 20-25 times per call (~1 us of a 3-8 us call, and a
// data race with any setenv in a multi-threaded graph).  The variables are now snapshotted into an immutable table when the library
// first needs one (and again on qdsp_hip_reload_env(), for tests and tuning scripts that change the environment in-process);
// a call site reads `knob(K_NAME, default)`: one atomic pointer load and an array access.  Old snapshots are never freed (a few
// hundred bytes per reload), so a reader that holds one stays valid.
#pragma once
#include <atomic>

namespace qk {

#define QDSP_HIP_KNOBS(X) \
    X(ANY_SMALL_CALL_TILES) \
    X(CHAN_ABL) \
    X(CHAN_NO_ST4) X(DECIM_SETTING) X(FFT1K_MAX_COUNT) X(FFT_ABL) X(FFT_DMA) \
    X(FFT_MIN_TAPS_DECIM) X(FFT_MIN_TAPS_REAL) \
    X(FFT_MIN_TAPS_SMALL) X(FFT_NT) X(FFT_PRUNE2_MAX_COUNT) X(FFT_WG_PER_CU) X(FIR_LAT_MAX_WORK) X(FIR_MODE) X(FIR_PICK) X(FORCE_ANY) \
    X(MF_BATCH_MIN_WORK) X(MF_DEPTH) X(MF_MIN_COUNT) X(MF_MIN_DECIM) \
    X(MF_TASK_MAX) X(NO_CHAN_BATCH) X(NO_DECIM_TABLE) X(NO_FFT1K) X(NO_FFT1K_REAL) X(NO_FIR_LAT) X(NO_FIR_TABLE) X(NO_LM) \
    X(NO_LM_SMALL_CALL_RULE) X(NO_MF) X(NO_MF_BATCH) X(NO_PFB) X(NO_RM) X(NO_RM_EXT) X(NO_SPLIT_UPLOAD) \
    X(NO_WIN) X(NT) X(PFB_MIN_COUNT) X(PFB_WG_PER_CU) X(R) X(RM_MIN_COUNT) X(RM_MIN_INTERP) \
    X(RM_WAVES_PER_SIMD) X(WIN_MAX_TAPS) X(WIN_R) 
enum Knob {
#define X(n) K_##n,
    QDSP_HIP_KNOBS(X)
#undef X
    K_COUNT
};

struct Knobs {
    int val[K_COUNT];
    unsigned char set[K_COUNT];
    unsigned long long fft_stamps;   // QDSP_HIP_FFT_STAMPS: device pointer for the diagnostic build of fir_fft_dmapk_kernel (scripts/stamp_fir_fft.py)
};

const Knobs* knobs_snapshot();       // the current table (built on first use)
void knobs_reload();                 // re-read the environment

inline int knob(Knob k, int dflt) {
    const Knobs* s = knobs_snapshot();
    return s->set[k] ? s->val[k] : dflt;
}

}  // namespace qk
