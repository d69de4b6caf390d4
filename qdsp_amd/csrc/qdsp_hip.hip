// qdsp_hip.hip -- host side of libqdsp_hip.so: engine state + the extern "C" boundary
// declared in include/qdsp_hip.h.  gfx950 only; no CPU fallback exists anywhere in this
// library: without a HIP device every entry point returns an error.



#include "engine.hip.h"

namespace qh {


Engine* as_engine(void* h, Kind k) {
    Engine* e = static_cast<Engine*>(h);
    if (!e || e->magic != kMagic || e->kind != k) return nullptr;
    return e;
}

// End-of-call wait of the host-pointer paths.  hipStreamSynchronize sleeps on an interrupt (~20 us to wake up),
// longer than the kernels of a reference-sized block take: poll the stream for a bounded time first
// (QDSP_HIP_SYNC_SPIN_US, default 200; 0 = always block).
hipError_t wait_stream(hipStream_t s) {
    static const int spin_us = 200;
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        do {
            const hipError_t q = hipStreamQuery(s);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
        } while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us));
    }
    return hipStreamSynchronize(s);
}

// The same for the library's shared stream: wait for THIS call's work only (an event recorded behind it), not for
// whatever the upstream blocks have queued for later blocks in the meantime.
hipError_t wait_event(hipEvent_t ev, hipStream_t s) {
    hipError_t rc = hipEventRecord(ev, s);
    if (rc != hipSuccess) return rc;
    static const int spin_us = 200;
    if (spin_us > 0) {
        const auto t0 = std::chrono::steady_clock::now();
        do {
            const hipError_t q = hipEventQuery(ev);
            if (q == hipSuccess) return hipSuccess;
            if (q != hipErrorNotReady) return q;
        } while (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(spin_us));
    }
    return hipEventSynchronize(ev);
}

// (Ordering the two ends of a link with events instead -- every handle on its own stream, two events per device
// buffer of the stream<T>, hipStreamWaitEvent before and hipEventRecord behind each kernel -- was built and
// measured: the cross-stream dependencies cost more than the serialisation they remove; SineSource -> VFO 23 -> 27 us
// per block, Splitter -> 4 / 16 x VFO 80 -> 120 / 386 -> 615 us.)
// One in-order stream per device for "pipelined" device-resident links (QDSP_HIP_LINK_PIPELINED): a producer
// launches into it and hands its block over without waiting; the consumer launches into the same stream, so
// the GPU runs the two in launch order -- which is the order the stream<T> protocol imposes on the host threads
// (the consumer reads a block only after the producer swapped it in, the producer reuses a buffer only after
// the consumer flushed it, and both launch before they swap / flush).
hipStream_t shared_stream(int device) {
    static std::mutex m;
    static hipStream_t tab[64] = {};
    std::lock_guard<std::mutex> lk(m);
    if (device < 0 || device >= 64) return nullptr;
    if (!tab[device]) {
        if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&tab[device], hipStreamNonBlocking) != hipSuccess) tab[device] = nullptr;
    }
    return tab[device];
}

long double turns_of(float re, float im) {
    // arg(phase_inc)/2pi in [0,1): the angle the reference's recursive phasor actually
    // advances by each sample is that of the ROUNDED float pair, not of the ideal theta.
    const long double two_pi = 6.283185307179586476925286766559005768L;
    long double t = atan2l((long double)im, (long double)re) / two_pi;
    t -= floorl(t);
    return t;
}
unsigned long long fx_of_turns(long double t) {
    t -= floorl(t);
    long double s = ldexpl(t, 64);
    if (s >= 18446744073709551615.0L) return 0ULL;
    return (unsigned long long)s;
}
void unit_of_fx(unsigned long long ph, long double mult, double* c, double* s) {
    // exp(j*2pi*frac(ph/2^64 * mult)) in long double, rounded to double
    const long double two_pi = 6.283185307179586476925286766559005768L;
    long double t = ldexpl((long double)ph, -64) * mult;
    t -= floorl(t);
    *c = (double)cosl(two_pi * t);
    *s = (double)sinl(two_pi * t);
}

// The same for the per-call NCO constants that only depend on the increment (step phasors, per-lane / per-row tables in
// the kernel arguments): memoised per thread -- a block's worker thread launches for one handle -- because at
// reference-sized calls 10-30 long-double sincos per call (1.5-4 us of host time) are as long as the kernel itself.
// 128 direct-mapped entries keyed by (increment, integer multiple); a retune simply misses.
void unit_of_fx_c(unsigned long long ph, long double mult, double* c, double* s) {
    struct Memo { unsigned long long ph; long long m; double c, s; bool ok; };
    static thread_local Memo memo[128];
    const long long m = (long long)mult;
    if ((long double)m != mult) { unit_of_fx(ph, mult, c, s); return; }
    const unsigned idx = (unsigned)(((ph * 0x9E3779B97F4A7C15ULL) ^ ((unsigned long long)m * 0xC2B2AE3D27D4EB4FULL)) >> 57);
    Memo& e = memo[idx];
    if (!(e.ok && e.ph == ph && e.m == m)) {
        unit_of_fx(ph, mult, &e.c, &e.s);
        e.ph = ph;
        e.m = m;
        e.ok = true;
    }
    *c = e.c;
    *s = e.s;
}

int64_t out_size(const Engine* e, int64_t count) {
    if (!e->has_filter) return count;
    return (count * e->L) / e->M;  // calcOutSize, resampling.h:95-97 (L = M = 1 for FIR)
}

// fir_core_kernel (de-interleaved tile): FIR and decimations up to 8.  From 9 on the general kernel with its
// staged-as-it-lies tile is 2-3x faster (scripts/tune_large_decim.py: M = 9..16, 31-255 taps: 0.12-0.38 ms per
// 2^26 samples against 0.26-0.54).
// per-call exceptions the measured decimator table may raise against the rule chain (process_dev sets Engine::auto_veto / auto_mode)
enum { VETO_WIN = 1, VETO_FFT1K = 2, VETO_PFB = 4, VETO_MF = 8 };
bool use_core(const Engine* e) { return e->L == 1 && e->M <= 8 && !qk::knob(qk::K_FORCE_ANY, 0); }

// decimators served by decim_win_kernel (kernels.hip.h): interp 1, short filters.  Outputs per lane and the
// tap limit from scripts/tune_win.py / tune_small.py (2^26 samples): chunks of M*R <= 10 samples are the sweet
// spot; past ~100-128 taps the overlap-save kernels take over (their pruned forms at M = 4, 8, 16 earlier).
int win_R(int M, int P) {
    const int r = qk::knob(qk::K_WIN_R, 0);      // experiments: 1, 2, 4 or 8 where instantiated
    if (r == 1 || r == 2 || r == 4 || r == 8) return r;
    if (M == 1) return 8;
    if (M <= 3) return 4;
    if (M == 4) return P > 96 ? 4 : 2;
    if (M <= 6) return 2;
    if (M == 8) return P > 63 ? 2 : 1;
    return 1;
}
int win_limit(int M) {
    static const int limit[17] = {0, 7, 150, 150, 160, 200, 256, 160, 200, 0, 192, 0, 224, 0, 0, 0, 256};
    return M >= 0 && M <= 16 ? limit[M] : 0;
}
bool use_win(const Engine* e) {
    if (!e->has_filter || e->L != 1) return false;
    const int M = e->M;
    if (!((M >= 1 && M <= 8) || M == 10 || M == 12 || M == 16)) return false;
    // where the overlap-save forms take over (per 2^26 samples they run 0.25 / 0.19 / 0.15 / 0.17 ms at decimation
    // 2 / 4 / 8 / 16 -- pruned inverse -- and 0.20-0.21 ms at every other decimation: full inverse, strided store)
    // ([1] = FIR<T> and equal-rate resamplers below the overlap-save threshold: 0.22 ms against 0.25 de-interleaved)
    const int max_taps = qk::knob(qk::K_WIN_MAX_TAPS, win_limit(M));
    return e->P <= max_taps && qk::knob(qk::K_NO_WIN, 0) == 0 && !(e->auto_veto & VETO_WIN);
}

// interp / decim pairs served by resamp_lm_kernel (kernels.hip.h)
bool use_lm(const Engine* e) {
    if (e->kind == KIND_FIR || !e->has_filter) return false;
    if (!(e->L == 2 || e->L == 3 || e->L == 4 || e->L == 5 || e->L == 10)) return false;
    return e->M >= 1 && e->M <= 8 && qk::knob(qk::K_NO_LM, 0) == 0;
}

// Shapes the MFMA decimator (mf_dec.hip.h) serves: complex data, interp 1, at most 16 taps per polyphase column (the 16
// rows of the A operand), decimation up to 128 (a tile of 16 rows is prefetched in registers).  Its time hardly depends
// on the tap count (0.21-0.25 ms per 2^27 samples from decimation 14 up, profiles/r02_tune_mf.md); below decimation 14
// the tiles get small and it only wins over the strided-window / general kernels from ~12 taps per unit of decimation,
// and the strided-window kernel keeps the short filters at decimation 16.
bool mf_plan(const Engine* e, int* KJ, int* QS, int* keep2) {
    // (ch == 1, round 4: PolyphaseResampler<float> -- decim_mfma_real_kernel, the same plan on float rows)
    if ((e->ch != 2 && e->ch != 1) || (e->ch == 1 && e->rotate) || e->L != 1 || !e->has_filter || e->kind == KIND_FIR) return false;
    const int M = e->M, P = e->P;
    if (M < qk::knob(qk::K_MF_MIN_DECIM, 9)) return false;
    // decimations 130-256 (even): the kernel runs rows of M / 2 samples -- the decimator by M / 2 with the same taps -- and
    // keeps every other output; twice the matrix work for the outputs that count, on a unit that has the room
    const int k2 = (M > 8 * qk::kMfMaxKJ && M <= 16 * qk::kMfMaxKJ && M % 2 == 0 && !0) ? 1 : 0;
    const int Mk = k2 ? M / 2 : M;
    if (Mk > 8 * qk::kMfMaxKJ) return false;
    const int Q = (P + Mk - 1) / Mk;                  // taps per column: one set of 16 rows of the A operand, or two
    if (Q > qk::kMfMaxQ || (Q > 16 && 0)) return false;
    if (Q > 16 && M < 12) return false;      // (two tap sets on rows of < 12 samples: 0.44 ms per 2^27 at decimation 10 against 0.36 overlap-save, round 3)
    if (M < 14 && P < 12 * M) return false;
    if (P < M) return false;                          // (fewer taps than the decimation: most of each row meets no tap, the general kernel is 5-16 % ahead)
    if (use_win(e) && P < 6 * M) return false;
    *KJ = (Mk + 7) / 8;
    *QS = Q > 16 ? 2 : 1;
    *keep2 = k2;
    return true;
}

int upload_taps(Engine* e, const float* taps, int ntaps) {
    // FIR: h[k] pairs with s[n - (ntaps-1) + k]  -> core with M=1, Q=ntaps, H=ntaps-1
    // resampler L==1: tapPhases[0][t] = taps[t]   -> core with M, Q=ceil(P/M), H=P
    // otherwise: [L][P] phase table, buildTapPhases (resampling.h:137-166)
    std::vector<float> host;
    if (use_core(e)) {
        const int K = e->P;
        const int Q = (K + e->M - 1) / e->M;
        host.assign((size_t)e->M * Q, 0.0f);
        for (int k = 0; k < K; k++) host[(size_t)(k % e->M) * Q + k / e->M] = taps[k];
    } else {
        host.assign((size_t)e->L * e->P, 0.0f);
        int cur = 0;
        for (int tap = 0; tap < e->P; tap++)
            for (int phase = 0; phase < e->L; phase++)
                host[(size_t)((e->L - 1) - phase) * e->P + tap] = (cur < ntaps) ? taps[cur++] : 0.0f;
    }
    if (e->d_taps) { HIPCHK(hipFree(e->d_taps)); e->d_taps = nullptr; }
    HIPCHK(hipMalloc(&e->d_taps, host.size() * sizeof(float)));
    HIPCHK(hipMemcpy(e->d_taps, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    if (e->d_taps_lm) { HIPCHK(hipFree(e->d_taps_lm)); e->d_taps_lm = nullptr; }
    if (use_win(e)) {
        // decim_win_kernel: hp[k + M*(R-1)] = h[k], zeros around (d_taps_lm doubles as its table)
        const int M = e->M, P = e->P, R = win_R(M, P), MR = M * R;
        const int nchunks = (M * (R - 1) + P + MR - 1) / MR;
        std::vector<float> w((size_t)nchunks * MR + (size_t)M * (R - 1) + 16, 0.0f);
        for (int k = 0; k < P; k++) w[(size_t)k + M * (R - 1)] = taps[k];
        HIPCHK(hipMalloc(&e->d_taps_lm, w.size() * sizeof(float)));
        HIPCHK(hipMemcpy(e->d_taps_lm, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (use_lm(e)) {
        // sub-filter c of resamp_lm_kernel: h_c = phases[(c*M) % L], stored branch-major [c][m][q] = h_c[q*M + m]
        const int L = e->L, M = e->M, P = e->P, Q = (P + M - 1) / M;
        std::vector<float> lm((size_t)L * M * Q, 0.0f);
        for (int c = 0; c < L; c++) {
            const float* hc = host.data() + (size_t)((c * M) % L) * P;
            for (int k = 0; k < P; k++) lm[((size_t)c * M + k % M) * Q + k / M] = hc[k];
        }
        // M == 1: the same taps transposed, [q][c], behind the first table
        const size_t n1 = lm.size();
        if (M == 1) {
            lm.resize(2 * n1);
            for (int c = 0; c < L; c++)
                for (int q = 0; q < Q; q++) lm[n1 + (size_t)q * L + c] = lm[(size_t)c * Q + q];
        }
        e->taps_lm_t_off = M == 1 ? n1 : 0;
        HIPCHK(hipMalloc(&e->d_taps_lm, lm.size() * sizeof(float)));
        HIPCHK(hipMemcpy(e->d_taps_lm, lm.data(), lm.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    // MFMA decimator (mf_dec.hip.h): A operand of set s, step (jj, a), lane l = tap row q = 16 s + l % 16, column 8 jj + 2 (l / 16) + a
    if (e->d_taps_mf) { HIPCHK(hipFree(e->d_taps_mf)); e->d_taps_mf = nullptr; }
    e->mf_KJ = 0;
    {
        int KJ = 0, QS = 1, keep2 = 0;
        if (mf_plan(e, &KJ, &QS, &keep2)) {
            const int Mk = keep2 ? e->M / 2 : e->M;
            std::vector<float> tk((size_t)QS * 2 * KJ * 64, 0.0f);
            for (int sset = 0; sset < QS; sset++)
                for (int jj = 0; jj < KJ; jj++)
                    for (int a = 0; a < 2; a++)
                        for (int l = 0; l < 64; l++) {
                            const int q = 16 * sset + l % 16, col = 8 * jj + 2 * (l / 16) + a, k = Mk * q + col;
                            if (col < Mk && k < e->P) tk[((size_t)(sset * KJ + jj) * 2 + a) * 64 + l] = taps[k];
                        }
            HIPCHK(hipMalloc(&e->d_taps_mf, tk.size() * sizeof(float)));
            HIPCHK(hipMemcpy(e->d_taps_mf, tk.data(), tk.size() * sizeof(float), hipMemcpyHostToDevice));
            e->mf_KJ = KJ;
            e->mf_QS = QS;
            e->mf_keep2 = keep2;
        }
    }
    // rational MFMA resampler (rm_resamp.hip.h): the banded period matrix W[i][c] = phases[(i M) % L][c - (i M) / L], cut
    // into blocks of 4 rows (16 blocks per MFMA step) with one band per block, in the lane order of the A operand.
    // Periods too short to hold their band in one row are merged J at a time (L' = J L, M' = J M: the same operator).
    if (e->d_taps_rm) { HIPCHK(hipFree(e->d_taps_rm)); e->d_taps_rm = nullptr; }
    e->rm_ngrp = 0;
    // Where it pays (profiles/r02_tune_rm.md): periods of at least 9 blocks (interp >= 33: 48 kHz <-> 44.1 kHz runs 1.25-1.5x
    // faster than through the general direct kernel), pure interpolators, and the decimating side of the small ratios
    // resamp_lm_kernel serves (decim >= 5, and 10/3: 1.2-2.2x at up to ~24 taps per phase).  Other short periods share a step
    // between period quads and run within +-15 % of the general kernel, which keeps them (QDSP_HIP_RM_MIN_INTERP lowers
    // the bar).
    bool rm_wanted = e->L >= qk::knob(qk::K_RM_MIN_INTERP, 33) || (e->M == 1 && e->L >= 6 && !use_lm(e));
    if (use_lm(e) && e->L >= 2 && !0)
        rm_wanted = rm_wanted || (e->M >= 5 && e->P <= (e->L == 10 ? 36 : 24)) || (e->L == 10 && e->M >= 3 && e->P <= 24);
    // Round 3 (scripts/sweep_rm_grid.py, profiles/r03_sweep_rm_grid.txt: 130 ratios x 8-32 taps per phase on 2^26-sample calls): the block
    // form is ahead of the general direct kernel on nearly every ratio from 14 taps per phase on (x 0.5-0.9 of its time; 72 of 79 ratios at
    // 14 taps per phase, 60 at 12 with losses up to x 1.25, at 8 anything from x 0.7 to x 1.4), and of resamp_lm_kernel on its decimating side up to 32 taps per phase (3/8: x 0.49, 5/8:
    // x 0.53, 4/7: x 0.61) and on 4/3, 5/3, 5/4 from 16 taps per phase (x 0.73-0.97); resamp_lm_kernel keeps decimations 1-2,
    // 2/3 and 3/4 (x 1.05-2.2).  These plans only serve chip-filling calls (rm_min_count): their reference-sized calls stay where
    // round 2's survey put them.
    e->rm_big_only = false;
    if (!rm_wanted && e->L >= 2 && !qk::knob(qk::K_NO_RM_EXT, 0)) {
        bool ext;
        if (use_lm(e))
            ext = (e->M >= 5 && e->P <= 36) || (e->L == 10 && e->M >= 3 && e->P <= 36) ||
                  ((e->M == 3 || e->M == 4) && e->L >= 4 && e->P >= 16 && e->P <= 32);      // (3/4: x 0.90-0.94 at 16 / 24 taps per phase, x 1.2 at 20: left alone)
        else      // (7/6, 9/7, 9/8 at 16 taps per phase: x 1.04-1.17, the only losers of that column)
            ext = e->M >= 2 && e->P >= 14 && !(e->P < 20 && ((e->L == 7 && e->M == 6) || (e->L == 9 && (e->M == 7 || e->M == 8))));
        if (ext) {
            rm_wanted = true;
            e->rm_big_only = true;
        }
    }
    // (ch == 1, round 4: PolyphaseResampler<float> -- resamp_mfma_real_kernel, the same plan on float tiles.  Against the real-data forms of the other
    // kernels -- scripts/sweep_real_rational.py forced, profiles/r04_real_rational.txt -- it wins on chip-filling calls at every tap count (147/160 with
    // 8 / 12 taps per phase: x 0.63 / 0.54 of their time) but for 33/32 below 14 taps per phase (x 1.05-1.10); small calls: rm_min_count)
    if (e->ch == 1 && rm_wanted && e->P < 14 && e->L >= 33 && e->L < 48) rm_wanted = false;
    if ((e->ch == 2 || (e->ch == 1 && !e->rotate)) && rm_wanted && e->has_filter && e->kind != KIND_FIR && !use_core(e) && e->M < (1 << 16)) {
        const int L0 = e->L, M0 = e->M, P = e->P;
        for (int J = 1; J <= 64 && !e->rm_ngrp; J++) {
            const int L = J * L0, M = J * M0, nblk = (L + 3) / 4;
            if (nblk > 16 * qk::kRmMaxGrp || 4 * M > 64 * qk::kRmNE) break;
            const int qpb = nblk <= 8 ? 16 / nblk : 1, ngrp = (nblk + 15) / 16;
            std::vector<int> c0((size_t)nblk);
            int KB = 1, reach = 0;
            for (int b = 0; b < nblk; b++) {
                const int i0 = 4 * b, i1 = (i0 + 3 < L - 1) ? i0 + 3 : L - 1;
                c0[b] = (int)(((long long)i0 * M0) / L0);
                const int need = (int)(((long long)i1 * M0) / L0) + P - c0[b];
                if (need > KB) KB = need;
            }
            if (KB > qk::kRmMaxKB) break;
            for (int b = 0; b < nblk; b++)
                if (c0[b] + KB > reach) reach = c0[b] + KB;
            const int ext = reach > M ? reach - M : 0;
            if (ext > M) continue;                     // a row must hold its band with one neighbour's head: merge more periods
            // used fraction of the 16 blocks of a step: below ~0.6 the general direct kernel is the faster one
            const double used = (double)(nblk * qpb) / (16.0 * ngrp);
            if (used < 0.6) continue;                  // (more merged periods may fill the steps better)
            int pitch = M + ext;
            pitch += (pitch & 1) ^ 1;                  // odd: the four periods of a quad start on different banks
            if (e->ch == 1) {      // (complex data: the same search changes nothing -- 0.4155 / 0.4378 / 0.4394 against 0.4139 / 0.4415 / 0.4431 ms, alternated)
                // Real data: the B operand of a step is one ds_read_b32 per lane -- 32-lane groups, bank = sample index mod 32 -- from sample
                // (period l % 4) * pitch + c0[block l / 4] + k.  PMC of resamp_mfma_real_kernel at the odd pitch: LDS 70 % busy, a third of it bank
                // conflicts (profiles/r04_pmc_resamp_mfma_real.json).  Of the 32 pitches from M + ext on, take the one whose lane groups meet the
                // fewest busy banks (the float tiles are half the size of the complex ones: the longer rows cost nothing that matters).
                int best_cost = 1 << 30;
                for (int cand = M + ext; cand < M + ext + 32; cand++) {
                    int cost = 0;
                    for (int g = 0; g < ngrp; g++)
                        for (int half = 0; half < 2; half++) {
                            int busy[32] = {0}, worst = 0;
                            for (int l = 32 * half; l < 32 * half + 32; l++) {
                                const int slot = l / 4;
                                const int b = ngrp > 1 || qpb == 1 ? 16 * g + slot : slot % nblk;
                                const int pq = ngrp > 1 || qpb == 1 ? 0 : slot / nblk;
                                const int sidx = ((4 * pq + l % 4) * cand + (b < nblk ? c0[b] : 0)) & 31;
                                if (++busy[sidx] > worst) worst = busy[sidx];
                            }
                            cost += worst;
                        }
                    if (cost < best_cost) { best_cost = cost; pitch = cand; }
                }
            }
            // period quads per tile: as many as the register prefetch (kRmNE samples per lane) holds, a multiple of qpb
            int G = (64 * qk::kRmNE - ext) / (4 * M);
            if (G > 16) G = 16;
            G -= G % qpb;
            if (G < 1 || qk::rm_lds_bytes(ngrp, KB, G, pitch, e->ch == 1) > 64 * 1024) continue;
            std::vector<float> tab((size_t)ngrp * KB * 64 + (size_t)2 * ngrp * 64, 0.0f);
            int* cbl = reinterpret_cast<int*>(tab.data() + (size_t)ngrp * KB * 64);
            int* meta = cbl + (size_t)ngrp * 64;
            for (int g = 0; g < ngrp; g++)
                for (int l = 0; l < 64; l++) {
                    const int slot = l / 4;            // block slot of the step
                    const int b = ngrp > 1 || qpb == 1 ? 16 * g + slot : slot % nblk;
                    const int pq = ngrp > 1 || qpb == 1 ? 0 : slot / nblk;
                    const bool live = b < nblk && pq < qpb;
                    cbl[g * 64 + l] = live ? c0[b] : 0;
                    meta[g * 64 + l] = live ? ((pq << 16) | b) : 0xffff;
                    const int i = 4 * b + l % 4;
                    if (!live || i >= L) continue;
                    const int off = (int)(((long long)i * M0) / L0), ph = (int)(((long long)i * M0) % L0);
                    for (int k = 0; k < KB; k++) {
                        const int col = c0[b] + k;
                        if (col >= off && col < off + P) tab[((size_t)g * KB + k) * 64 + l] = host[(size_t)ph * P + (col - off)];
                    }
                }
            HIPCHK(hipMalloc(&e->d_taps_rm, tab.size() * sizeof(float)));
            HIPCHK(hipMemcpy(e->d_taps_rm, tab.data(), tab.size() * sizeof(float), hipMemcpyHostToDevice));
            e->rm_ngrp = ngrp;
            e->rm_KB = KB;
            e->rm_ext = ext;
            e->rm_pitch = pitch;
            e->rm_G = G;
            e->rm_J = J;
            e->rm_qpb = qpb;
        }
    }
    return 0;
}

// (Re)configure the filter part.  History keeps its newest samples across a resize.
int configure(Engine* e, const float* taps, int ntaps, int interp, int decim) {
    if (!taps || ntaps <= 0 || interp <= 0 || decim <= 0) return QDSP_HIP_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    const int oldH = e->H;
    e->L = interp;
    e->M = decim;
    e->ntaps = ntaps;
    e->P = (ntaps + interp - 1) / interp;
    const int newH = (e->kind == KIND_FIR) ? ntaps - 1 : e->P;
    int rc = upload_taps(e, taps, ntaps);
    if (rc) return rc;
    e->taps_host.assign(taps, taps + ntaps);
    e->fft_ntaps = -1;
    e->f1k_ntaps = -1;
    e->pfb_ntaps = -1;
    const size_t bytes = (size_t)(newH > 0 ? newH : 1) * e->ch * sizeof(float);
    float* nh[2] = {nullptr, nullptr};
    for (int i = 0; i < 2; i++) {
        HIPCHK(hipMalloc(&nh[i], bytes));
        HIPCHK(hipMemset(nh[i], 0, bytes));
    }
    if (e->d_hist[0] && oldH > 0 && newH > 0) {
        const int keep = oldH < newH ? oldH : newH;
        HIPCHK(hipMemcpy(nh[0] + (size_t)(newH - keep) * e->ch,
                         e->d_hist[e->cur] + (size_t)(oldH - keep) * e->ch,
                         (size_t)keep * e->ch * sizeof(float), hipMemcpyDeviceToDevice));
    }
    for (int i = 0; i < 2; i++)
        if (e->d_hist[i]) HIPCHK(hipFree(e->d_hist[i]));
    e->d_hist[0] = nh[0];
    e->d_hist[1] = nh[1];
    e->cur = 0;
    e->raw_valid = false;
    e->H = newH;
    e->hist_cap = newH;
    return 0;
}

void set_inc_now(Engine* e, float re, float im) {
    e->raw_valid = false;   // (a retune: the rotated history stays the truth, see launch_fft)
    e->inc_re = re;
    e->inc_im = im;
    e->dturns = turns_of(re, im);
    e->dphase = fx_of_turns(e->dturns);
    e->gm1 = (float)(hypotl((long double)re, (long double)im) - 1.0L);
}

// setters: stage the increment; the worker thread applies it between two calls, never in the middle of one
void set_inc(Engine* e, float re, float im) {
    unsigned int a, b;
    memcpy(&a, &re, 4);
    memcpy(&b, &im, 4);
    e->pending_inc.store(((unsigned long long)a << 32) | b, std::memory_order_release);
}
void apply_pending_inc(Engine* e) {
    const unsigned long long v = e->pending_inc.exchange(0ULL, std::memory_order_acq_rel);
    if (!v) return;
    const unsigned int a = (unsigned int)(v >> 32), b = (unsigned int)v;
    float re, im;
    memcpy(&re, &a, 4);
    memcpy(&im, &b, 4);
    set_inc_now(e, re, im);
}

int ensure_io(Engine* e, int max_block) {
    if (max_block <= 0) return 0;
    HIPCHK(hipSetDevice(e->device));
    const size_t in_bytes = (size_t)max_block * e->ch * sizeof(float);
    const size_t oc = (size_t)out_size(e, max_block) + 1;
    if (!e->d_in || max_block > e->max_block) {
        if (e->d_in) HIPCHK(hipFree(e->d_in));
        HIPCHK(hipMalloc(&e->d_in, in_bytes));
        e->max_block = max_block;
    }
    if (!e->d_out || oc > e->out_cap) {
        if (e->d_out) HIPCHK(hipFree(e->d_out));
        HIPCHK(hipMalloc(&e->d_out, oc * e->ch * sizeof(float)));
        e->out_cap = oc;
    }
    return 0;
}

int create(void** h, Kind kind, int device, int ch, bool rotate, bool has_filter, int max_block) {
    if (!h) return QDSP_HIP_EINVAL;
    *h = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return QDSP_HIP_ENODEV;
    if (device < 0 || device >= ndev) return QDSP_HIP_ENODEV;
    HIPCHK(hipSetDevice(device));
    Engine* e = new (std::nothrow) Engine();
    if (!e) return QDSP_HIP_ENOMEM;
    e->kind = kind;
    e->device = device;
    e->ch = ch;
    e->rotate = rotate;
    e->has_filter = has_filter;
    e->R = qk::knob(qk::K_R, 0);
    e->NT = qk::knob(qk::K_NT, 0);
    hipError_t err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipEventCreate(&e->ev0);
    if (err == hipSuccess) err = hipEventCreate(&e->ev1);
    if (err != hipSuccess) {
        // nothing created so far may outlive the failed handle
        if (e->ev1) (void)hipEventDestroy(e->ev1);
        if (e->ev0) (void)hipEventDestroy(e->ev0);
        if (e->stream) (void)hipStreamDestroy(e->stream);
        delete e;
        return -(int)err;
    }
    e->max_block = 0;
    (void)max_block;
    *h = e;
    return 0;
}

void destroy(Engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->device);
    (void)hipDeviceSynchronize();
    if (e->d_taps) (void)hipFree(e->d_taps);
    if (e->d_taps_lm) (void)hipFree(e->d_taps_lm);
    for (int i = 0; i < 2; i++)
        if (e->d_hist_raw[i]) (void)hipFree(e->d_hist_raw[i]);
    if (e->d_nco_tab) (void)hipFree(e->d_nco_tab);
    if (e->d_fft_H) (void)hipFree(e->d_fft_H);
    if (e->d_fft_TA) (void)hipFree(e->d_fft_TA);
    if (e->d_fft_TB) (void)hipFree(e->d_fft_TB);
    if (e->d_f1k_H) (void)hipFree(e->d_f1k_H);
    if (e->d_f1k_T) (void)hipFree(e->d_f1k_T);
    if (e->d_pfb) (void)hipFree(e->d_pfb);
    if (e->d_taps_mf) (void)hipFree(e->d_taps_mf);
    if (e->d_taps_rm) (void)hipFree(e->d_taps_rm);
    for (int i = 0; i < 2; i++)
        if (e->d_hist[i]) (void)hipFree(e->d_hist[i]);
    if (e->d_in) (void)hipFree(e->d_in);
    if (e->d_out) (void)hipFree(e->d_out);
    if (e->ev0) (void)hipEventDestroy(e->ev0);
    if (e->ev1) (void)hipEventDestroy(e->ev1);
    if (e->ev_up) (void)hipEventDestroy(e->ev_up);
    if (e->ev_kernel) (void)hipEventDestroy(e->ev_kernel);
    if (e->up_stream) (void)hipStreamDestroy(e->up_stream);
    if (e->stream) (void)hipStreamDestroy(e->stream);
    e->magic = 0;
    delete e;
}

// ---- launches --------------------------------------------------------------------------

int nco_tables(Engine* e, long long S, int NT, int na, const double2** tab) {
    if (e->d_nco_tab && e->nco_key_dphase == e->dphase && e->nco_key_S == S && e->nco_key_NT == NT && e->nco_key_na >= na) {
        *tab = e->d_nco_tab;
        return 0;
    }
    if (na < 64) na = 64;
    std::vector<double2> h((size_t)256 + na + NT);
    for (int b = 0; b < 256; b++) unit_of_fx_c(e->dphase, (long double)b * (long double)S, &h[b].x, &h[b].y);
    for (int a2 = 0; a2 < na; a2++) unit_of_fx_c(e->dphase, 256.0L * (long double)a2 * (long double)S, &h[256 + a2].x, &h[256 + a2].y);
    for (int t = 0; t < NT; t++) unit_of_fx_c(e->dphase, (long double)t, &h[256 + na + t].x, &h[256 + na + t].y);
    HIPCHK(hipDeviceSynchronize());       // (rare: nothing in flight may still read the old tables)
    if (e->d_nco_tab) HIPCHK(hipFree(e->d_nco_tab));
    e->d_nco_tab = nullptr;
    HIPCHK(hipMalloc(&e->d_nco_tab, h.size() * sizeof(double2)));
    HIPCHK(hipMemcpy(e->d_nco_tab, h.data(), h.size() * sizeof(double2), hipMemcpyHostToDevice));
    e->nco_key_dphase = e->dphase;
    e->nco_key_S = S;
    e->nco_key_NT = NT;
    e->nco_key_na = na;
    *tab = e->d_nco_tab;
    return 0;
}

// NCO constants of stage_tile (kernels.hip.h) for a kernel staging with NT lanes; with the tile geometry (S samples
// between tiles, `first` = stream position staged by lane 0 of tile 0, ntiles) also the tile_phasor tables



// host copy of qk::slot (must match kernels.hip.h)
inline int qk_slot_host(int R, int v) { return (R % 2 == 0) ? v + v / R : v; }

template <int CH, int R, int NT, bool ROT, int MT>
int launch_core_t(Engine* e, qk::CoreArgs& a, hipStream_t s) {
    constexpr int TILE = NT * R;
    const int V = TILE + a.Q;
    int sb = qk_slot_host(R, V - 1) + 1;
    // Spread the M branch bases over the banks for the de-interleaving LDS writes.
    if (a.M > 1) {
        const int want = (a.M >= 16) ? 1 : 16 / a.M;
        while ((sb % 16) != (want % 16)) sb++;
    }
    a.sb = sb;
    const size_t lds = (size_t)a.M * sb * CH * sizeof(float);
    if (lds > (size_t)kMaxDynLds) return QDSP_HIP_EINVAL;
    a.nblocks = (int)((a.nout + TILE - 1) / TILE);
    a.vec = ((uintptr_t)a.in & 15) == 0;
    if (ROT) {
        { int rcn = fill_stage_rot(e, a, NT, (long long)TILE * a.M, -(long long)a.H, a.nblocks); if (rcn) return rcn; }
        unit_of_fx_c(a.dphase, 1.0L, &a.rot_one.x, &a.rot_one.y);
        unit_of_fx_c(a.dphase, (long double)(2 * NT), &a.rot_2nt.x, &a.rot_2nt.y);
    }
    hipLaunchKernelGGL((qk::fir_core_kernel<CH, R, NT, ROT, MT>), dim3(a.nblocks + 1), dim3(NT), lds, s, a);
    HIPCHK(hipGetLastError());
    e->last.name = "fir_core_kernel";
    e->last.grid = a.nblocks + 1;
    e->last.block = NT;
    e->last.lds = (int)lds;
    return 0;
}

template <int CH, bool ROT> int launch_core(Engine* e, qk::CoreArgs& a, hipStream_t s) {
    // Geometry: (R outputs per lane, NT lanes).  FIR (M == 1) is VALU-bound: a long
    // register window amortises the LDS read per tap.  Decimators stage M*TILE inputs per
    // tile, so TILE shrinks with M to stay inside the LDS budget.
    int R = e->R, NT = e->NT;
    if (R == 0 || NT == 0) {
        if (a.M == 1) { R = 8; NT = 256; }
        else if (a.M <= 2) { R = 8; NT = 128; }
        else if (a.M <= 8) { R = 4; NT = 128; }
        else { R = 4; NT = 64; }
    }
    const int mt = (a.M == 1 || a.M == 2 || a.M == 4 || a.M == 8) ? a.M : 0;
#define QK_CASE(r, nt, m) \
    if (R == r && NT == nt && mt == m) return launch_core_t<CH, r, nt, ROT, m>(e, a, s);
#define QK_GEOM(r, nt) QK_CASE(r, nt, 0) QK_CASE(r, nt, 2) QK_CASE(r, nt, 4) QK_CASE(r, nt, 8)
    QK_CASE(8, 256, 1)
    QK_CASE(16, 256, 1)
    QK_CASE(4, 256, 1)
    QK_GEOM(8, 256)
    QK_GEOM(4, 256)
    QK_GEOM(8, 128)
    QK_GEOM(4, 128)
    QK_GEOM(4, 64)
#undef QK_GEOM
#undef QK_CASE
    return QDSP_HIP_EINVAL;
}

template <int CH, bool ROT> int launch_win(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s) {
    qk::WinArgs a;
    memset(&a, 0, sizeof(a));
    a.in = d_in;
    a.out = d_out;
    a.hist = e->d_hist[e->cur];
    a.hist_next = e->d_hist[e->cur ^ 1];
    a.taps = e->d_taps_lm;
    a.count = count;
    a.nout = nout;
    a.P = e->H;                  // window start / history length: P for the resampler, ntaps-1 for FIR<T>
    a.ntaps = e->P;
    a.phase0 = e->phase;
    a.dphase = e->dphase;
    a.gm1 = e->volk_gain ? e->gm1 : 0.0f;
    constexpr int NT = 256;
    auto go = [&](auto Mc, auto Rc) -> int {
        constexpr int M = decltype(Mc)::value, R = decltype(Rc)::value, MR = M * R;
        constexpr int TILE = NT * R;
        a.nchunks = (M * (R - 1) + e->P + MR - 1) / MR;
        const int U = TILE * M + a.nchunks * MR;
        const size_t lds = (size_t)(U + U / MR + 1) * CH * sizeof(float);
        if (lds > (size_t)kMaxDynLds) return QDSP_HIP_EINVAL;
        a.nblocks = (int)((nout + TILE - 1) / TILE);
        { int rcn = fill_stage_rot(e, a, NT, (long long)TILE * M, -(long long)a.P, a.nblocks); if (rcn) return rcn; }
        hipLaunchKernelGGL((qk::decim_win_kernel<CH, M, R, NT, ROT>), dim3(a.nblocks + 1), dim3(NT), lds, s, a);
        HIPCHK(hipGetLastError());
        e->last.name = "decim_win_kernel";
        e->last.grid = a.nblocks + 1;
        e->last.block = NT;
        e->last.lds = (int)lds;
        return 0;
    };
    using std::integral_constant;
    const int R = win_R(e->M, e->P);
#define QK_WIN(m, r) if (e->M == m && R == r) return go(integral_constant<int, m>{}, integral_constant<int, r>{});
    QK_WIN(1, 4) QK_WIN(1, 8)
    QK_WIN(2, 2) QK_WIN(2, 4) QK_WIN(2, 8)
    QK_WIN(3, 2) QK_WIN(3, 4)
    QK_WIN(4, 1) QK_WIN(4, 2) QK_WIN(4, 4)
    QK_WIN(5, 1) QK_WIN(5, 2) QK_WIN(5, 4)
    QK_WIN(6, 1) QK_WIN(6, 2)
    QK_WIN(7, 1)
    QK_WIN(8, 1) QK_WIN(8, 2)
    QK_WIN(10, 1) QK_WIN(12, 1) QK_WIN(16, 1)
#undef QK_WIN
    return QDSP_HIP_EINVAL;
}

template <int CH, bool ROT> int launch_lm(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s) {
    qk::LmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = d_in;
    a.out = d_out;
    a.hist = e->d_hist[e->cur];
    a.hist_next = e->d_hist[e->cur ^ 1];
    a.taps = e->d_taps_lm;
    a.taps_t = e->d_taps_lm + e->taps_lm_t_off;
    a.count = count;
    a.nout = nout;
    a.M = e->M;
    a.P = e->P;
    a.Q = (e->P + e->M - 1) / e->M;
    for (int c = 0; c < e->L; c++) a.e[c] = (c * e->M) / e->L;
    a.phase0 = e->phase;
    a.dphase = e->dphase;
    a.gm1 = e->volk_gain ? e->gm1 : 0.0f;
    constexpr int NT = 128;
    // R*L accumulators per lane: 10 .. 30 complex values
    auto go = [&](auto Rc, auto Lc) -> int {
        constexpr int R = decltype(Rc)::value, LL = decltype(Lc)::value;
        constexpr int TJ = NT * R;
        const int V = TJ + a.Q + 1;
        int sb = V;
        if ((sb & 15) == 0) sb += 1;                    // keep the M branch bases off one bank for the de-interleaving writes
        a.sb = sb;
        size_t lds = (size_t)a.M * sb * CH * sizeof(float);
        const size_t lds_out = (size_t)TJ * LL * CH * sizeof(float);      // the tile's outputs pass through LDS too
        if (lds < lds_out) lds = lds_out;
        if (lds > (size_t)kMaxDynLds) return QDSP_HIP_EINVAL;
        a.nblocks = (int)((nout + (long long)TJ * LL - 1) / ((long long)TJ * LL));
        { int rcn = fill_stage_rot(e, a, NT, (long long)TJ * a.M, -(long long)a.P, a.nblocks); if (rcn) return rcn; }
        hipLaunchKernelGGL((qk::resamp_lm_kernel<CH, R, NT, ROT, LL>), dim3(a.nblocks + 1), dim3(NT), lds, s, a);
        HIPCHK(hipGetLastError());
        e->last.name = "resamp_lm_kernel";
        e->last.grid = a.nblocks + 1;
        e->last.block = NT;
        e->last.lds = (int)lds;
        return 0;
    };
    using std::integral_constant;
    switch (e->L) {
        case 2: return go(integral_constant<int, 5>{}, integral_constant<int, 2>{});
        case 3: return go(integral_constant<int, 5>{}, integral_constant<int, 3>{});
        case 4: return go(integral_constant<int, 3>{}, integral_constant<int, 4>{});
        case 5: return go(integral_constant<int, 3>{}, integral_constant<int, 5>{});
        case 10: return go(integral_constant<int, 3>{}, integral_constant<int, 10>{});
        default: return QDSP_HIP_EINVAL;
    }
}

// Tile plan of resamp_any_kernel: phase-table pitch and bytes in LDS (0: table stays in memory), padded layout,
// outputs per tile (0: the taps of one phase do not fit) and the LDS elements a tile stages.
AnyPlan any_plan(int L, int M, int P, int ch, long long nout) {
    constexpr int NT = 256;
    AnyPlan p;
    // the phase table rides in LDS when it leaves at least half of the budget to the samples
    p.Pp = (P + 3) & ~3;
    if (((p.Pp >> 2) & 1) == 0) p.Pp += 4;
    const long long tap_bytes = (long long)L * p.Pp * (long long)sizeof(float);
    const bool lt = tap_bytes <= kMaxDynLds / 2 && 0 == 0;
    p.tap_bytes = lt ? (int)tap_bytes : 0;
    // Tile = as many outputs as keep the staged input span inside what is left of the LDS budget.
    const long long max_elems = (kMaxDynLds - p.tap_bytes) / (ch * (int)sizeof(float));
    long long tile = (long long)8 * NT;
    // interp 1 with a decimation that is a multiple of 4: lane windows M samples apart share LDS banks (4-way and
    // worse) -> one pad element per M samples
    p.pad = L == 1 && (M & 3) == 0 && M <= 65536 && 0 == 0;
    auto span_of = [&](long long t) {
        const long long sp = ((t - 1) * M) / L + P + 2;
        return p.pad ? sp + sp / M + 1 : sp;
    };
    // interp 1 with a tile of at most a quarter of the workgroup: NT / tile lanes share an output's taps (NT partial
    // sums behind the samples).  Two lanes per output (tile 128) measured no better than one: the extra barrier
    // and LDS round trip cost what the halved tap loop saves (M = 50, 201 taps: 0.162 vs 0.145 ms).
    constexpr int kSplitTile = NT / 4;
    const bool ks_ok = L == 1 && P >= 64 && 0 == 0;
    auto need = [&](long long t) { return span_of(t) + ((ks_ok && t <= kSplitTile) ? NT : 0); };
    while (tile > 1 && need(tile) > max_elems) tile /= 2;
    // reference-sized calls (nout known): 2048 outputs per tile leave a 1e6-sample block of a 24/125 audio resampler on 94
    // workgroups and a 16 384-sample one on 2 (10.3 us per call whatever the size); one to four outputs per lane spread it
    if (nout >= 0) {
        const long long want = qk::knob(qk::K_ANY_SMALL_CALL_TILES, 512);   // (256 .. 4096 measured: 147/160 at 1e6 samples 9.5 / 9.1 / 10.0 / 11.2 us)
        while (tile > NT && (nout + tile - 1) / tile < want) tile /= 2;
        // large decimations (taps split over the lanes of an output): down to 16 outputs per tile = 16 lanes per output, as long
        // as a lane keeps 16 taps -- the VFO's 401 taps / 50 on a 1e6-sample block: 100 dependent MACs per lane -> 25
        if (ks_ok && want > 0) {
            if (tile > kSplitTile && (nout + kSplitTile - 1) / kSplitTile <= 768 && P >= 64) tile = kSplitTile;
            // (... and as long as the halved tiles still fit one round of the chip: the NCO variant keeps 3 workgroups per CU
            // resident, and a 769th workgroup waits for a whole round -- 401-513 taps / 40-64 on 1e6 samples: 9.4-9.5 us with
            // 783-978 tiles against 7.3-7.4 with 392-490)
            const long long tmin = 16;
            while (tile > tmin && tile <= kSplitTile && (nout + tile - 1) / tile < want && (nout + tile / 2 - 1) / (tile / 2) <= 768 &&
                   P / (2 * NT / tile) >= 16)
                tile /= 2;
        }
    }
    // a half-workgroup tile with a long tap loop: the quarter tile with four lanes per output is faster (M = 50,
    // 401 taps: 0.24 -> 0.17 ms) unless it stages too little per lane (M = 32: 9 samples in batches of 8)
    if (ks_ok && tile == 2 * kSplitTile && P >= 192 && M >= 40) tile = kSplitTile;
    p.tile = need(tile) > max_elems ? 0 : tile;
    p.span = span_of(tile);
    p.ks_lanes = p.ks_shift = p.ks_chunk = 0;
    if (ks_ok && p.tile >= 1 && p.tile <= kSplitTile && (p.tile & (p.tile - 1)) == 0) {
        p.ks_lanes = (int)(NT / p.tile);
        while ((1 << p.ks_shift) < p.tile) p.ks_shift++;
        p.ks_chunk = (((P + p.ks_lanes - 1) / p.ks_lanes) + 3) & ~3;
    }
    return p;
}

// Tile geometry of resamp_any_kernel into `a` (a.L / a.M / a.P / a.nout set by the caller); returns the dynamic LDS
// bytes, 0 if the taps of one phase do not fit.
size_t fill_any_geometry(qk::AnyArgs& a, int ch, bool* lt, bool* pad, int nchan) {
    constexpr int NT = 256;
    const AnyPlan pl = any_plan(a.L, a.M, a.P, ch, a.nout * nchan);   // (a batch launch runs nchan x nblocks tiles)
    if (pl.tile == 0) return 0;  // taps per phase beyond LDS
    *lt = pl.tap_bytes != 0;
    *pad = pl.pad;
    const long long tile = pl.tile;
    a.Pp = pl.Pp;
    a.tap_bytes = pl.tap_bytes;
    a.pad_inv = pl.pad ? (unsigned)((1ULL << 32) / (unsigned)a.M) + 1u : 0u;
    a.tile = (int)tile;
    a.ks_lanes = pl.ks_lanes;
    a.ks_shift = pl.ks_shift;
    a.ks_chunk = pl.ks_chunk;
    a.ks_red = (int)pl.span;
    a.nblocks = (int)((a.nout + tile - 1) / tile);
    a.step_d = (int)(((long long)NT * a.M) / a.L);
    a.step_p = (int)(((long long)NT * a.M) % a.L);
    return (size_t)a.tap_bytes + (size_t)(pl.span + (pl.ks_lanes ? NT : 0)) * ch * sizeof(float);
}

template <int CH, bool ROT> int launch_any(Engine* e, qk::AnyArgs& a, hipStream_t s) {
    constexpr int NT = 256;
    bool lt = false, pad = false;
    const size_t lds = fill_any_geometry(a, CH, &lt, &pad);
    if (lds == 0) return QDSP_HIP_EINVAL;  // taps per phase beyond LDS
    // persistent workgroups, 8 per CU (measured on the M = 50, 401-tap VFO, 32 KB of LDS each: 5 per CU -- what is
    // resident at once -- 0.41 ms per 2^27 samples, 8 .. 64 per CU 0.345-0.358)
    int nwg = 256 * 8;
    if (nwg > a.nblocks) nwg = a.nblocks;
    if (nwg < 1) nwg = 1;
    a.nwg = nwg;
    // full persistent grids only: a contiguous tile range per XCD (kernels.hip.h)
    a.xcd_tiles = (nwg >= 2048 && (nwg & 7) == 0 && !0) ? (a.nblocks + 7) / 8 : 0;
    fill_stage_rot(a, NT);
    if (pad) {
        if (lt) hipLaunchKernelGGL((qk::resamp_any_kernel<CH, NT, ROT, true, true>), dim3(nwg + 1), dim3(NT), lds, s, a);
        else hipLaunchKernelGGL((qk::resamp_any_kernel<CH, NT, ROT, false, true>), dim3(nwg + 1), dim3(NT), lds, s, a);
    } else {
        if (lt) hipLaunchKernelGGL((qk::resamp_any_kernel<CH, NT, ROT, true, false>), dim3(nwg + 1), dim3(NT), lds, s, a);
        else hipLaunchKernelGGL((qk::resamp_any_kernel<CH, NT, ROT, false, false>), dim3(nwg + 1), dim3(NT), lds, s, a);
    }
    HIPCHK(hipGetLastError());
    e->last.name = "resamp_any_kernel";
    e->last.grid = nwg + 1;
    e->last.block = NT;
    e->last.lds = (int)lds;
    return 0;
}

// ---- overlap-save FFT FIR -------------------------------------------------------------------
constexpr int kFftMaxTaps = 2049;   // keeps >= 2048 valid outputs per 4096-point block

// DEC the overlap-save kernel would run with for this engine, 0 if it cannot: 1 = full inverse
// (the FIR, and any other integer decimation through a strided store), 2/4/8/16 = pruned inverse.
int fft_dec(const Engine* e) {
    if (e->L != 1 || e->ntaps < 2 || e->ntaps > kFftMaxTaps) return 0;
    if (e->ch == 1) {
        // real data: two real segments per complex transform (full inverse; decimators keep every M-th output)
        if (e->kind == KIND_FIR) return 1;
        if (e->kind == KIND_DECIM) return 1;
        return 0;
    }
    if (e->kind == KIND_FIR) return 1;
    if (e->kind == KIND_DECIM || e->kind == KIND_VFO) {
        if (e->M == 2 || e->M == 4 || e->M == 8 || e->M == 16) return e->M;
        return 1;   // any other decimation, 1 included (the pure xlating FIR / a resampler at equal rates)
    }
    return 0;
}

// explicit settings (set_mode, QDSP_HIP_FIR_MODE) outrank the table's per-call mode
int mode_of(const Engine* e) {
    if (e->fir_mode) return e->fir_mode;
    const int k = qk::knob(qk::K_FIR_MODE, 0);
    return k ? k : e->auto_mode;
}

// ---- FIR<complex_t>, AUTO: which kernel family serves a call of `count` samples with `ntaps` taps -------------------------------------------
// Round 4: this one choice is DATA, not an if-chain.  scripts/sweep_fir_table.py times the four families on a grid of
// call sizes x tap counts (each forced with QDSP_HIP_FIR_PICK), scripts/gen_dispatch_table.py turns the committed sweep
// (profiles/r04_sweep_fir_table.txt) into dispatch_table.inc -- one byte per cell: the fastest family -- tests/test_capi_cpu.py checks that the
// table is the sweep's, tests/test_gpu_dispatch.py re-measures a fixed sample of cells and fails when the default is > 10 % behind an alternative.
// A call takes the nearest cell (log2 of the count, log of the taps); a family that cannot serve the shape (structural limits in the *_eligible
// predicates) hands the call back to the rule chain.
enum FirPick { PICK_NONE = 0, PICK_LAT = 1, PICK_CORE = 2, PICK_FFT1K = 3, PICK_FFT4K = 4 };
#include "dispatch_table.inc"
// The switch settings the decimator sweep tries (scripts/sweep_decim_table.py lists them in the same order): what each one vetoes / which mode it sets.
constexpr int kDecimSettingVeto[16] = {0, VETO_WIN, VETO_FFT1K, VETO_WIN | VETO_FFT1K, 0, 0, VETO_PFB, VETO_PFB, VETO_MF, 0, 0, 0, 0, 0, 0, 0};
constexpr int kDecimSettingMode[16] = {0, 0, 0, 0, 1, 2, 0, 2, 0, 0, 0, 0, 0, 0, 0, 0};
#include "decim_table.inc"
int decim_table_setting(int rot, int M, int ntaps, int64_t count) {   // rot: the table's class -- 0 complex decimator, 1 fused VFO, 2 real data
    if (qk::knob(qk::K_NO_DECIM_TABLE, 0) || count <= 0 || rot >= kDecimTabClasses) return 0;
    // the decimation's row: below 10 the swept decimation itself or none (the kernel families change from one small decimation to the next: 6 on
    // 5's row lost 40 % where the rules were right); from 10 on -- general direct kernel, overlap-save, MFMA decimator at every decimation -- the
    // swept decimation nearest on a log scale (12 -> 10, 40 -> 32, 80 -> 64), inside the swept range.  The settings only take kernels AWAY from
    // the rule chain or name the family, so a neighbour's setting is always valid; tests/test_gpu_dispatch.py holds unswept decimations to
    // the same 10 % as swept ones.
    if (M < kDecimTabM[0] || M > kDecimTabM[kDecimTabMs - 1] || ntaps < kDecimTabTaps[0] / 2) return 0;
    int mi = -1;
    for (int i = 0; i < kDecimTabMs; i++)
        if (kDecimTabM[i] == M) mi = i;
    for (int i = 1; i < kDecimTabMs && mi < 0; i++)
        if (kDecimTabM[i - 1] >= 10 && M > kDecimTabM[i - 1] && M < kDecimTabM[i])
            mi = (long long)M * M >= (long long)kDecimTabM[i - 1] * kDecimTabM[i] ? i : i - 1;      // (the geometric mean is the border)
    if (mi < 0) return 0;
    int lg = 0;
    while ((int64_t(1) << (lg + 1)) <= count) lg++;
    if (count - (int64_t(1) << lg) > (int64_t(1) << lg) * 0.41421356) lg++;
    int row = lg - kDecimTabLog2Min;
    row = row < 0 ? 0 : row >= kDecimTabRows ? kDecimTabRows - 1 : row;
    int col = 0;
    for (int c = 1; c < kDecimTabCols; c++)
        if ((double)ntaps * ntaps >= (double)kDecimTabTaps[c - 1] * kDecimTabTaps[c]) col = c;
    return kDecimTab[rot][mi][row][col];
}
int fir_table_pick(int64_t count, int ntaps) {
    if (count <= 0 || ntaps < kFirPickTaps[0]) return PICK_NONE;
    int lg = 0;
    while ((int64_t(1) << (lg + 1)) <= count) lg++;                   // floor(log2 count)
    if (count - (int64_t(1) << lg) > (int64_t(1) << lg) * 0.41421356) lg++;   // nearest power of two on a log scale (sqrt 2)
    int row = lg - kFirPickLog2Min;
    row = row < 0 ? 0 : row >= kFirPickRows ? kFirPickRows - 1 : row;
    int col = 0;
    for (int c = 1; c < kFirPickCols; c++)                            // nearest grid point on a log scale: the geometric mean is the border
        if ((double)ntaps * ntaps >= (double)kFirPickTaps[c - 1] * kFirPickTaps[c]) col = c;
    return kFirPick[row][col];
}

// Large decimations (the VFO's usual job: 2.4 Msps -> 48 kHz is M = 50) with the few taps per output such
// filters have: the general direct kernel streams the input once and does P/M MACs per input sample, while the
// overlap-save form pays a full 4096-point transform pair whatever M (0.20-0.22 ms per 2^26 samples, 0.25-0.28
// with the fused NCO).  Measured (scripts/tune_large_decim.py, M = 9..250, 31..1001 taps): the direct form takes
// 0.12-0.17 ms up to ~9 taps per input-sample-of-decimation and ~320 taps (tiles shrink with M: LDS holds
// tile * M samples), and the NCO costs it 0.015 ms instead of 0.055; 2-way bank conflicts (M = 2 mod 4) move
// the crossover down.
bool any_direct_wins(const Engine* e) {
    if (e->L != 1 || e->ch != 2 || 0) return false;
    const AnyPlan pl = any_plan(1, e->M, e->P, e->ch);
    if (pl.tile == 0) return false;
    // tiles of 64 outputs and fewer (M >= 64 or so) split every output's taps over 4-16 lanes: 0.12-0.26 ms up to
    // 2001 taps at ~8 taps per unit of decimation (the reference VFO's own design rule) against 0.22-0.41
    if (pl.ks_lanes) return e->P <= (e->rotate ? 12 : 10) * e->M;
    const int per_m = (e->M & 3) == 2 ? (e->rotate ? 10 : 7) : (e->rotate ? 15 : 9);
    const int max_taps = e->rotate ? 512 : 320;
    return e->P <= per_m * e->M && e->P <= max_taps;
}

bool fft1k_eligible(const Engine* e, int64_t count);

bool fft_eligible(const Engine* e, int64_t count) {
    if (!fft_dec(e)) return false;
    int mode = mode_of(e);
    if (mode == 1) return false;
    if (mode == 2) return true;
    if (e->auto_pick) return e->auto_pick == PICK_FFT1K || e->auto_pick == PICK_FFT4K;
    // auto: calls big enough to fill the chip with 4096-point segments, and filters past the measured
    // crossover of the two forms
    // real data (two segments per transform, 305 Gs/s whatever the taps at 2^26 samples): the direct form
    // moves half the bytes per sample and stays ahead to ~96 taps (445 Gs/s at 63, 164 at 256); a real
    // decimator keeps 1/M of a full inverse, so the direct form wins until ~32 taps per branch
    // (decimate-by-8, 256 taps: 324 vs 321 Gs/s)
    int min_taps;
    if (e->ch == 1) {
        // Chip-filling calls (round 3, profiles/r03_sweep_real.txt, 2^27 real samples): the overlap-save form costs 0.23-0.28 ms at any
        // length; against it the direct form stays ahead to 64 taps (FIR), 128-192 taps at decimation 2-5, 384 at decimation 8, the
        // strided-window kernel to its own limits at decimation 10 / 12 / 16 -- and nothing else serves the decimations in between:
        // round 2's 32 taps per unit of decimation left decimate-by-10 with 256 taps on the general kernel at 0.56 ms (now 0.26).
        const bool big = count >= (1 << 22) && !0;
        min_taps = qk::knob(qk::K_FFT_MIN_TAPS_REAL, big && e->M == 1 ? 64 : 96);
        if (e->M > 1) {
            int need = 32 * e->M;
            if (big) {
                static const int t[9] = {0, 0, 144, 128, 192, 176, 224, 224, 384};
                // (swept: decimations 1-5, 8, 10, 12, 16 -- profiles/r03_sweep_real.txt; past 16 the round-2 rule stays: ADVICE round 3)
                need = e->M <= 8 ? t[e->M] : win_limit(e->M) ? win_limit(e->M) + 1 : e->M <= 16 ? 128 : 32 * e->M;
            }
            if (min_taps < need) min_taps = need;
        }
    } else if (e->M == 1 && e->kind != KIND_FIR) {
        // equal-rate resampler / pure xlating FIR: full inverse + per-element store (0.52 ms per 2^27 samples)
        // against the tile-per-block direct form (0.51 ms at 7 taps, 0.63 at 63)
        min_taps = 24;
    } else if (e->M == 1) {
        // FIR: the overlap-save kernel (a copy-speed 4.8 TB/s whatever the taps) beats the tile-per-block
        // direct form from 8 taps on (2^26 samples: 0.222 vs 0.256 ms at 7 taps, 0.225 vs 0.394 at 127)
        min_taps = 8;
        // reference-sized calls (<= 1e6 samples, stream.h:7) are latency-bound: one 4096-point segment takes ~9 us
        // whatever the taps, the direct form 4.7 / 5.7 us at 31 / 63 taps (8.3 at 1e6 samples) and 11-16 us at 256
        // (round 2: with one-wave 1024-point segments -- fft1k_fir.hip -- the small calls cross over at ~24 taps:
        // 31 taps x 1e6 samples 6.8 us against 7.2, 95 taps x 262144 5.7 against 6.9; profiles/r02_tune_fft1k.txt)
        if (count < (1 << 21) && min_taps < 96) min_taps = qk::knob(qk::K_FFT_MIN_TAPS_SMALL, fft1k_eligible(e, count) ? 24 : 96);
    } else {
        if (e->M >= 9 && !use_win(e) && any_direct_wins(e)) return false;
        // decimators (scripts/tune_small.py, profiles/r01_tune_small.txt): the direct form slows down with the
        // decimation (LDS-capacity-bound de-interleaved tiles): M = 2 / 4 / 5 it wins to ~110 taps
        // (0.20 / 0.16 / 0.18 ms vs 0.24 / 0.18 / 0.22), from M = 7 the overlap-save form wins at any length
        // (M = 8: 0.16 vs 0.25-0.29 ms; M = 16: 0.17 vs 0.45-0.49 ms)
        const int dflt = e->M >= 7 ? 2 : 112;
        min_taps = qk::knob(qk::K_FFT_MIN_TAPS_DECIM, dflt);
    }
    // (below 2^16 samples a 4096-point segment per workgroup leaves most of the chip idle; one-wave segments go down to 2^14)
    // (... and further where the alternative is fir_core_kernel on a long filter: 256-400 taps at decimation 3 / 8, 2048-16 000
    // samples: 10-13 us against 5.6-5.9; the FIR itself has fir_lat_kernel there)
    // Long filters outside that kernel's range (more than 769 taps): the direct kernels cost 4-7.5e-6 us per tap and
    // sample (1500 taps x 4096 samples: 49 us; 1000 taps at decimation 7: 29 us up to 65 534 samples), one 4096-point
    // segment 9.5-12 us whatever it holds: from 2^21 tap-samples on the segment wins (the decimators' direct kernel does not
    // get cheaper with the decimation: its time follows the taps a lane walks through)
    int64_t min_count = 1 << 16;
    if (fft1k_eligible(e, count)) {
        // (FIR<complex_t> has fir_lat_kernel below 2^24 tap-samples; FIR<float> only fir_core_kernel: 3.4 us + 4e-6 per tap and sample)
        if (e->kind != KIND_FIR) min_count = 64;
        else if (e->ch == 2) min_count = e->ntaps > 320 ? 64 : 1 << 14;
        else min_count = (1 << 19) / e->ntaps < 1024 ? 1024 : (1 << 19) / e->ntaps;
    }
    if (min_count == (1 << 16) && e->ntaps >= 256) {
        const int64_t by_work = (1 << 21) / e->ntaps;
        min_count = by_work < 1024 ? 1024 : by_work;
    }
    return e->ntaps >= min_taps && count >= ((int)min_count);
}

// Spectrum of a short sequence zero-padded to F = 2^m points, FP64 radix-2 (twiddles from sincosl, rounded once): what the
// overlap-save kernels multiply by.  The direct long-double DFT this replaces cost F x taps x87 operations -- 4-8 ms per
// handle at 4096 points and 256 taps, on creation and on every retune of a fused VFO; this is ~0.1 ms, and its error
// (1e-16 x log2 F) stays nine orders below the FP32 rounding of the result.
void host_spectrum(const std::vector<long double>& gr, const std::vector<long double>& gi, int F, std::vector<double>& re, std::vector<double>& im) {
    re.assign((size_t)F, 0.0);
    im.assign((size_t)F, 0.0);
    int bits = 0;
    while ((1 << bits) < F) bits++;
    for (size_t j = 0; j < gr.size() && j < (size_t)F; j++) {
        unsigned r = 0;
        for (int b = 0; b < bits; b++) r |= ((j >> b) & 1u) << (bits - 1 - b);
        re[r] = (double)gr[j];
        im[r] = (double)gi[j];
    }
    const long double two_pi = 6.283185307179586476925286766559005768L;
    std::vector<double> wr((size_t)F / 2), wi((size_t)F / 2);
    for (int i = 0; i < F / 2; i++) {
        wr[i] = (double)cosl(two_pi * (long double)i / F);
        wi[i] = (double)(-sinl(two_pi * (long double)i / F));
    }
    for (int len = 2; len <= F; len <<= 1) {
        const int half = len >> 1, step = F / len;
        for (int base = 0; base < F; base += len)
            for (int k = 0; k < half; k++) {
                const double cr = wr[(size_t)k * step], ci = wi[(size_t)k * step];
                const double xr = re[base + k + half], xi = im[base + k + half];
                const double tr = xr * cr - xi * ci, ti = xr * ci + xi * cr;
                re[base + k + half] = re[base + k] - tr;
                im[base + k + half] = im[base + k] - ti;
                re[base + k] += tr;
                im[base + k] += ti;
            }
    }
}

int fft_prepare(Engine* e) {
    // fused VFO: the spectrum is that of taps[k] * exp(j k dphase) (fft_fir.hip.h), so it follows the NCO
    const unsigned long long key_dphase = e->rotate ? e->dphase : 0;
    if (e->fft_ntaps == e->ntaps && e->fft_tw_ready && e->fft_dphase == key_dphase) return 0;
    constexpr int F = qk::kFftN;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    std::vector<long double> cs, sn;
    if (!e->fft_tw_ready) {   // (the twiddle tables below: once per handle -- the flag, not a pointer: it is set only after every
                              // allocation and upload of the first-time block has succeeded)
        cs.resize(F);
        sn.resize(F);
        for (int i = 0; i < F; i++) {
            cs[i] = cosl(two_pi * (long double)i / F);
            sn[i] = sinl(two_pi * (long double)i / F);
        }
    }
    const int N = e->ntaps;
    // g[j] = taps[N-1-j]:  c[p] = sum_j g[j] s[p-j] = sum_k taps[k] s[p-(N-1)+k];
    // Hf[k] = sum_j g[j] exp(-j 2pi jk/F) / F
    std::vector<float2> Hp(F), TA(256 * 16), TB(16 * 16);
    std::vector<long double> gr(N), gi(N);
    for (int j = 0; j < N; j++) {
        const long double h = (long double)e->taps_host[N - 1 - j];
        long double c = 1.0L, sn_ = 0.0L;
        if (e->rotate) {
            const long double tt = ldexpl((long double)e->dphase, -64) * (long double)(N - 1 - j);
            c = cosl(two_pi * (tt - floorl(tt)));
            sn_ = sinl(two_pi * (tt - floorl(tt)));
        }
        gr[j] = h * c;
        gi[j] = h * sn_;
    }
    std::vector<double> sre, sim;
    host_spectrum(gr, gi, F, sre, sim);
    for (int k = 0; k < F; k++) {
        const int k0 = k & 15, k1 = (k >> 4) & 15, k2 = k >> 8;
        Hp[(k0 * 16 + k1) * 16 + k2] = make_float2((float)(sre[k] / F), (float)(sim[k] / F));
    }
    if (!e->fft_tw_ready) {
        for (int t = 0; t < 256; t++)
            for (int k = 0; k < 16; k++) TA[t * 16 + k] = make_float2((float)cs[(t * k) % F], (float)(-sn[(t * k) % F]));
        for (int lo = 0; lo < 16; lo++)
            for (int k = 0; k < 16; k++) TB[lo * 16 + k] = make_float2((float)cs[(16 * lo * k) % F], (float)(-sn[(16 * lo * k) % F]));
        if (!e->d_fft_H) HIPCHK(hipMalloc(&e->d_fft_H, sizeof(float2) * F));
        if (!e->d_fft_TA) HIPCHK(hipMalloc(&e->d_fft_TA, sizeof(float2) * 256 * 16));
        if (!e->d_fft_TB) HIPCHK(hipMalloc(&e->d_fft_TB, sizeof(float2) * 16 * 16));
        HIPCHK(hipMemcpy(e->d_fft_TA, TA.data(), sizeof(float2) * TA.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(e->d_fft_TB, TB.data(), sizeof(float2) * TB.size(), hipMemcpyHostToDevice));
        e->fft_tw_ready = true;
    }
    HIPCHK(hipDeviceSynchronize());   // (retune / new taps: nothing in flight may still read the old spectrum)
    HIPCHK(hipMemcpy(e->d_fft_H, Hp.data(), sizeof(float2) * F, hipMemcpyHostToDevice));
    e->fft_ntaps = N;
    e->fft_dphase = key_dphase;
    return 0;
}

// ---- overlap-save on 1024-point segments, one wave each (fft1k_fir.hip) --------------------------------
// Reference-sized calls: a lone 4096-point segment takes its workgroup ~12 us whatever else runs (8 barriers), so a
// call that cannot fill the chip four workgroups deep is better off as 4x as many independent one-wave segments.
// Bounds: complex data, interp 1, taps up to half a segment; calls from the first size the 4096-point form is
// picked for up to QDSP_HIP_FFT1K_MAX_COUNT (measured crossover, scripts/tune_call_size.py).
bool fft1k_eligible(const Engine* e, int64_t count) {
    if (e->L != 1 || e->ntaps < 2 || e->ntaps > 769) return false;   // (769 taps: a quarter of every segment is new)
    if (e->kind != KIND_FIR && e->kind != KIND_DECIM && e->kind != KIND_VFO) return false;
    if (e->ch != 2 && (e->kind == KIND_VFO || e->rotate || qk::knob(qk::K_NO_FFT1K_REAL, 0))) return false;   // real data: two real segments per wave
    const int mode = mode_of(e);
    if (mode != 0 || qk::knob(qk::K_NO_FFT1K, 0) || (e->auto_veto & VETO_FFT1K)) return false;
    if (e->auto_pick) return e->auto_pick == PICK_FFT1K;
    const int forced = qk::knob(qk::K_FFT1K_MAX_COUNT, -1);
    if (forced >= 0) return count <= forced;
    // measured crossovers against the 4096-point kernels (scripts/tune_fft1k.py, profiles/r02_tune_fft1k.txt): the
    // overlap grows with the taps (1024 - ntaps + 1 new points per segment), decimations 2 / 4 / 8 / 16 have the
    // pruned inverse on the other side, the fused NCO costs this form 32 more complex products per lane
    if (e->ch == 1) {
        // real data (two real segments per wave): the other side is fir_fft_kernel<1, false, REAL> with its full inverse at every
        // decimation -- 256 taps: FIR 59.6 against 69.9 us at 2^25 samples, decimate-by-8 115.8 against 152.2 at 2^26
        if (e->ntaps > 513) return count <= (1 << 20);
        if (e->kind == KIND_FIR) return count <= (e->ntaps <= 128 ? 1 << 27 : e->ntaps <= 288 ? 1 << 25 : e->ntaps <= 416 ? 1 << 24 : 1 << 23);
        return count <= (e->ntaps <= 288 ? 1 << 27 : e->ntaps <= 416 ? 1 << 26 : 1 << 25);
    }
    const bool pruned = e->kind != KIND_FIR && (e->M == 2 || e->M == 4 || e->M == 8 || e->M == 16);
    int64_t lim = e->ntaps <= 128 ? 6 << 20 : e->ntaps <= 288 ? 4 << 20 : e->ntaps <= 416 ? 3 << 20 : e->ntaps <= 513 ? 2 << 20 : pruned ? 1 << 18 : 1 << 19;
    if (pruned && e->ntaps <= 513) lim = e->ntaps <= 288 ? 3 << 20 : 3 << 19;
    if (e->rotate && (pruned || e->ntaps > 288) && e->ntaps <= 513) lim = 3 << 19;
    return count <= lim;
}

int fft1k_prepare(Engine* e) {
    const unsigned long long key_dphase = e->rotate ? e->dphase : 0;
    if (e->f1k_ntaps == e->ntaps && e->f1k_tw_ready && e->f1k_dphase == key_dphase) return 0;
    constexpr int F = qk::kFft1kN;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    std::vector<long double> cs, sn;
    if (!e->f1k_tw_ready) {   // (the twiddle tables below: once per handle; see fft_prepare for the flag)
        cs.resize(F);
        sn.resize(F);
        for (int i = 0; i < F; i++) {
            cs[i] = cosl(two_pi * (long double)i / F);
            sn[i] = sinl(two_pi * (long double)i / F);
        }
    }
    const int N = e->ntaps;
    // g[j] = taps[N-1-j] (x exp(j (N-1-j) dphase) for the fused VFO), Hf[k] = sum_j g[j] exp(-j 2pi jk/F) / F: as fft_prepare
    std::vector<long double> gr(N), gi(N);
    for (int j = 0; j < N; j++) {
        const long double h = (long double)e->taps_host[N - 1 - j];
        long double c = 1.0L, sn_ = 0.0L;
        if (e->rotate) {
            const long double tt = ldexpl((long double)e->dphase, -64) * (long double)(N - 1 - j);
            c = cosl(two_pi * (tt - floorl(tt)));
            sn_ = sinl(two_pi * (tt - floorl(tt)));
        }
        gr[j] = h * c;
        gi[j] = h * sn_;
    }
    std::vector<float2> Hp(F), T(64 * 16 + 4 * 16);
    std::vector<double> sre, sim;
    host_spectrum(gr, gi, F, sre, sim);
    for (int k = 0; k < F; k++) {
        const double re = sre[k], im = sim[k];
        // bin k = ka + 16 (4 g + s) + 256 kb0 sits with lane (ka << 2) | g, entry 4 s + kb0; tables are entry-major
        // ([entry][lane]: a wave's load of one entry is 512 contiguous bytes)
        const int ka = k & 15, kb1 = (k >> 4) & 15, kb0 = k >> 8;
        Hp[(4 * (kb1 & 3) + kb0) * 64 + ((ka << 2) | (kb1 >> 2))] = make_float2((float)(re / F), (float)(im / F));
    }
    if (!e->f1k_tw_ready) {
        for (int l = 0; l < 64; l++)
            for (int k = 0; k < 16; k++) T[k * 64 + l] = make_float2((float)cs[(l * k) % F], (float)(-sn[(l * k) % F]));
        for (int j = 0; j < 4; j++)
            for (int k = 0; k < 16; k++) T[1024 + k * 4 + j] = make_float2((float)cs[(16 * j * k) % F], (float)(-sn[(16 * j * k) % F]));
        if (!e->d_f1k_H) HIPCHK(hipMalloc(&e->d_f1k_H, sizeof(float2) * F));
        if (!e->d_f1k_T) HIPCHK(hipMalloc(&e->d_f1k_T, sizeof(float2) * T.size()));
        HIPCHK(hipMemcpy(e->d_f1k_T, T.data(), sizeof(float2) * T.size(), hipMemcpyHostToDevice));
        e->f1k_tw_ready = true;
    }
    HIPCHK(hipDeviceSynchronize());   // (retune / new taps: nothing in flight may still read the old spectrum)
    HIPCHK(hipMemcpy(e->d_f1k_H, Hp.data(), sizeof(float2) * F, hipMemcpyHostToDevice));
    e->f1k_ntaps = N;
    e->f1k_dphase = key_dphase;
    return 0;
}

int raw_history(Engine* e, hipStream_t s, const float2** hist, float2** hist_raw_next);

int launch_fft1k(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s) {
    int rc = fft1k_prepare(e);
    if (rc) return rc;
    qk::FftArgs a;
    memset(&a, 0, sizeof(a));
    a.in = static_cast<const float2*>(d_in);
    a.out = static_cast<float2*>(d_out);
    a.hist = reinterpret_cast<const float2*>(e->d_hist[e->cur]);
    a.hist_keep = a.hist;
    a.hist_next = reinterpret_cast<float2*>(e->d_hist[e->cur ^ 1]);
    rc = raw_history(e, s, &a.hist, &a.hist_raw_next);
    if (rc) return rc;
    a.Hf = e->d_f1k_H;
    a.TA = e->d_f1k_T;
    a.TB = e->d_f1k_T + 1024;
    a.count = count;
    a.nout = nout;
    a.H = e->H;
    a.dec = 1;
    a.rot = e->rotate ? 1 : 0;
    a.decm = 1;
    a.ov = e->ntaps - 1;
    if (e->kind != KIND_FIR) {
        // resampler / VFO: y[n'] sits at stream position n' M - 1: segments start one sample early
        a.decm = e->M;
        a.decm_inv = (e->M >= 2 && (unsigned long long)(e->M + qk::kFft1kN) * (unsigned long long)e->M < (1ULL << 32)) ? (unsigned)((1ULL << 32) / (unsigned)e->M) + 1u : 0u;
        a.strided = 1;
        a.m_shift = -1;
        for (int sh = 0; sh <= 6; sh++)
            if (e->M == (1 << sh)) a.m_shift = sh;
        a.seg_shift = a.ov + 1;
        a.L = qk::kFft1kN - a.ov;
        a.nblocks = (int)((count + 1 + a.L - 1) / a.L);
    } else {
        a.seg_shift = a.ov;
        a.L = qk::kFft1kN - a.ov;
        a.nblocks = (int)((count + a.L - 1) / a.L);
    }
    if (e->ch == 1) {   // real data: one wave = a PAIR of real segments (re / im of one transform)
        a.real2 = 1;
        a.nblocks = (a.nblocks + 1) / 2;
    }
    a.nwg = a.nblocks;
    if (a.rot) {
        a.phase_in0 = e->phase;
        a.phase0 = e->phase - (unsigned long long)(e->ntaps - 1) * e->dphase;
        a.dphase = e->dphase;
        a.gm1 = e->volk_gain ? e->gm1 : 0.0f;
        if (!e->wtab1k_ok || e->wtab1k_dphase != e->dphase) {
            for (int i = 0; i < 16; i++) {
                double c, sn;
                unit_of_fx_c(e->dphase, (long double)(64 * i), &c, &sn);
                e->wtab1k[i] = make_float2((float)c, (float)sn);
            }
            e->wtab1k_dphase = e->dphase;
            e->wtab1k_ok = true;
        }
        memcpy(a.wtab, e->wtab1k, sizeof(a.wtab));
    }
    rc = qk::launch_fir_fft1k(a, s);
    if (rc) return rc;
    e->raw_valid = e->rotate && e->H > 0;   // (the caller flips cur: the raw hand-over then sits at d_hist_raw[cur])
    e->last.name = "fir_fft1k_kernel";
    e->last.grid = a.nblocks + (a.H + 63) / 64;
    e->last.block = 64;
    e->last.lds = (int)(16 * qk::kFft1kPitch * sizeof(float2));
    return 0;
}

int launch_xlate_inc(Engine* e, const void* d_in, int64_t count, void* d_out, unsigned long long phase0, unsigned long long dphase, float gm1,
                     hipStream_t s);

// ---- polyphase overlap-save decimate-by-8 (pfb_dec.hip) ---------------------------------------------
// Serves PolyphaseResampler<complex_t> / the fused VFO with interp 1, decim 8 on calls big enough to fill the chip
// with one segment per wave (8 waves per CU: measured crossover against the per-segment fir_fft_kernel<8> in
// scripts/tune_call_size.py).
// (round 3: decimate by 4 as TWO output phases over the same eight columns, pfb_dec.hip: taps per column counted with the odd
// outputs' four-sample delay)
inline int pfb_phases(const Engine* e) { return e->M == 4 ? 2 : 1; }
inline int pfb_Q(const Engine* e) { return (e->ntaps + e->M * (pfb_phases(e) - 1) + qk::kPfbD - 1) / qk::kPfbD; }
bool pfb_eligible(const Engine* e, int64_t count) {
    // (ch == 1, round 4: PolyphaseResampler<float> at decimation 8 / 4 -- pfb_dec8_real_kernel / pfb_dec4_real_kernel, two real segments per set of transforms)
    const bool real8 = e->ch == 1 && !e->rotate && (e->M == qk::kPfbD || e->M == 4) && e->kind == KIND_DECIM;
    if ((e->ch != 2 && !real8) || e->L != 1 || (e->M != qk::kPfbD && e->M != 4) || e->ntaps < 2) return false;
    if (e->kind != KIND_DECIM && e->kind != KIND_VFO) return false;
    if (pfb_Q(e) > qk::kPfbMaxQ) return false;
    if (qk::knob(qk::K_NO_PFB, 0) || (e->auto_veto & VETO_PFB)) return false;
    // measured crossover (scripts/tune_pfb_threshold.py, 256 taps): a lone segment takes a wave ~7 us (15 us per call
    // with the table load) where fir_fft_kernel<8> needs 8 us, so the per-segment kernels keep the reference-sized
    // calls; from 2^23 samples (decimator) / 2^24 (fused VFO) on this form is ahead, 1.2x at 2^27
    // (never below one segment: the kernel's prefetch reads whole 4096-sample segments from a clamped in-range start)
    if (count < qk::kPfbSeg) return false;
    // (decimate by 4, profiles/r03_tune_pfb4.txt: level with fir_fft_dec_kernel<4> at 2^25 samples, 0.166 against 0.181 ms at 2^26, 0.312 against 0.330 at 2^27)
    return count >= (int64_t)qk::knob(qk::K_PFB_MIN_COUNT, e->M == 4 ? 1 << 26 : (e->rotate || real8) ? 1 << 24 : 1 << 23);      // (real data: pairs of segments)
}

int pfb_prepare(Engine* e) {
    const unsigned long long key_dphase = e->rotate ? e->dphase : 0;
    if (e->pfb_ntaps == e->ntaps && e->pfb_M == e->M && e->d_pfb && e->pfb_dphase == key_dphase) return 0;
    constexpr int F = qk::kPfbF, D = qk::kPfbD, R = qk::kPfbRow;
    const long double two_pi = 6.283185307179586476925286766559005768L;
    const int N = e->ntaps, PH = pfb_phases(e), Q = pfb_Q(e);
    // g[j] = taps[N-1-j] (x exp(j (N-1-j) dphase) for the fused VFO: the mixer folded into the taps, fft_fir.hip.h)
    std::vector<long double> gr((size_t)Q * D, 0.0L), gi((size_t)Q * D, 0.0L);
    for (int j = 0; j < N; j++) {
        const long double h = (long double)e->taps_host[N - 1 - j];
        long double c = 1.0L, sn = 0.0L;
        if (e->rotate) {
            const long double tt = ldexpl((long double)e->dphase, -64) * (long double)(N - 1 - j);
            c = cosl(two_pi * (tt - floorl(tt)));
            sn = sinl(two_pi * (tt - floorl(tt)));
        }
        gr[j] = h * c;
        gi[j] = h * sn;
    }
    std::vector<long double> cs(F), ss(F);
    for (int i = 0; i < F; i++) { cs[i] = cosl(two_pi * i / F); ss[i] = sinl(two_pi * i / F); }
    std::vector<float2> tab((size_t)qk::kPfbTableElems + (size_t)(PH - 1) * qk::kPfbG1Elems, make_float2(0.0f, 0.0f));
    // G: row (k0*64 + kq*8 + c), element kg: spectrum of column c's filter gamma_c[q] = g[8q + 7 - c] at bin k0 + 8 kq + 64 kg, / 512
    for (int c = 0; c < D; c++) {
        std::vector<long double> cr((size_t)Q), ci((size_t)Q);
        for (int q = 0; q < Q; q++) {
            cr[q] = gr[D * q + (D - 1 - c)];
            ci[q] = gi[D * q + (D - 1 - c)];
        }
        std::vector<double> sre, sim;
        host_spectrum(cr, ci, F, sre, sim);      // (FP64 radix-2 FFT: see fft_prepare)
        for (int k = 0; k < F; k++) {
            const int k0 = k & 7, kq = (k >> 3) & 7, kg = k >> 6;
            tab[(size_t)((k0 * 64 + kq * 8 + c) * R + kg)] = make_float2((float)(sre[k] / F), (float)(sim[k] / F));
        }
    }
    if (PH == 2) {
        // G1: the odd outputs y4[2n' - 1]: window end 8n' - 5, i.e. the same taps four samples later, g_1 = 0 0 0 0 ++ g; the NCO phase of
        // that output is four samples behind the even one's (x exp(-j 4 dphase), folded in here).  Rows of 8 values, 16-byte chunk i of row
        // (k0, kq, c) stored at chunk i ^ ((kq*8 + c) >> 2 & 3): pfb_dec.hip reads it back the same way
        long double rc = 1.0L, rs = 0.0L;
        if (e->rotate) {
            const long double tt = ldexpl((long double)e->dphase, -64) * 4.0L;
            rc = cosl(two_pi * (tt - floorl(tt)));
            rs = -sinl(two_pi * (tt - floorl(tt)));
        }
        float2* G1 = tab.data() + qk::kPfbTableElems;
        for (int c = 0; c < D; c++) {
            std::vector<long double> cr((size_t)Q, 0.0L), ci((size_t)Q, 0.0L);
            for (int q = 0; q < Q; q++) {
                const int j = D * q + (D - 1 - c) - 4;       // index into g
                if (j < 0 || j >= N) continue;
                cr[q] = gr[j] * rc - gi[j] * rs;
                ci[q] = gr[j] * rs + gi[j] * rc;
            }
            std::vector<double> sre, sim;
            host_spectrum(cr, ci, F, sre, sim);
            for (int k = 0; k < F; k++) {
                const int k0 = k & 7, kq = (k >> 3) & 7, kg = k >> 6;
                const int row = k0 * 64 + kq * 8 + c, swz = ((kq * 8 + c) >> 2) & 3;
                G1[(size_t)row * 8 + 2 * ((kg >> 1) ^ swz) + (kg & 1)] = make_float2((float)(sre[k] / F), (float)(sim[k] / F));
            }
        }
    }
    float2* TW = tab.data() + 512 * R;     // row (k0*8 + kq), element g': W512^(g' (k0 + 8 kq))
    for (int k0 = 0; k0 < 8; k0++)
        for (int kq = 0; kq < 8; kq++)
            for (int g = 0; g < 8; g++) {
                const int idx = (g * (k0 + 8 * kq)) % F;
                TW[(k0 * 8 + kq) * R + g] = make_float2((float)cs[idx], (float)(-ss[idx]));
            }
    float2* TI1 = TW + 64 * R;             // row m, element alpha: exp(+j 2pi m alpha / 512)
    for (int m = 0; m < 64; m++)
        for (int al = 0; al < 8; al++) TI1[m * R + al] = make_float2((float)cs[(m * al) % F], (float)ss[(m * al) % F]);
    float2* TI2 = TI1 + 64 * R;            // row m0, element b0: exp(+j 2pi m0 b0 / 64)
    for (int m0 = 0; m0 < 8; m0++)
        for (int b0 = 0; b0 < 8; b0++) TI2[m0 * R + b0] = make_float2((float)cs[(8 * m0 * b0) % F], (float)ss[(8 * m0 * b0) % F]);
    float2* EL = TI2 + 8 * R;              // exp(j 2pi 8 l dphase): the NCO over a lane's element offset (fused VFO)
    for (int l = 0; l < 64; l++) {
        double c = 1.0, sn = 0.0;
        if (e->rotate) unit_of_fx_c(e->dphase, (long double)(8 * l), &c, &sn);
        EL[l] = make_float2((float)c, (float)sn);
    }
    if (!e->d_pfb) HIPCHK(hipMalloc(&e->d_pfb, ((size_t)qk::kPfbTableElems + qk::kPfbG1Elems) * sizeof(float2)));
    HIPCHK(hipDeviceSynchronize());   // (retune / new taps: nothing in flight may still read the old tables)
    HIPCHK(hipMemcpy(e->d_pfb, tab.data(), tab.size() * sizeof(float2), hipMemcpyHostToDevice));
    e->pfb_ntaps = N;
    e->pfb_M = e->M;
    e->pfb_dphase = key_dphase;
    return 0;
}

int launch_xlate_inc(Engine* e, const void* d_in, int64_t count, void* d_out, unsigned long long phase0, unsigned long long dphase, float gm1,
                     hipStream_t s);

// The overlap-save kernels of the fused VFO filter RAW samples while the handle keeps its history rotated (the direct
// kernels and *_set_history_dev use that form; a call may switch form with its size): the H samples are de-rotated once
// per call into a side buffer -- exp(-j phi(g)), g = -H .. -1 -- unless the previous overlap-save call left them there.
int raw_history(Engine* e, hipStream_t s, const float2** hist, float2** hist_raw_next) {
    *hist_raw_next = nullptr;
    if (!(e->rotate && e->H > 0)) return 0;
    if (!e->d_hist_raw[0] || e->hist_raw_cap < e->H) {
        for (int i = 0; i < 2; i++) {
            if (e->d_hist_raw[i]) HIPCHK(hipFree(e->d_hist_raw[i]));
            e->d_hist_raw[i] = nullptr;
            HIPCHK(hipMalloc(&e->d_hist_raw[i], (size_t)e->H * sizeof(float2)));
        }
        e->hist_raw_cap = e->H;
        e->raw_valid = false;
    }
    if (!e->raw_valid || 0) {
        const unsigned long long ph_first = e->phase - (unsigned long long)e->H * e->dphase;   // phase of history sample 0
        const Launch keep = e->last;
        const int rc = launch_xlate_inc(e, e->d_hist[e->cur], e->H, e->d_hist_raw[e->cur], 0ULL - ph_first, 0ULL - e->dphase, 0.0f, s);
        e->last = keep;
        if (rc) return rc;
    }
    *hist = reinterpret_cast<const float2*>(e->d_hist_raw[e->cur]);
    *hist_raw_next = reinterpret_cast<float2*>(e->d_hist_raw[e->cur ^ 1]);
    return 0;
}

int launch_pfb(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s) {
    int rc = pfb_prepare(e);
    if (rc) return rc;
    qk::PfbArgs a;
    memset(&a, 0, sizeof(a));
    a.in = static_cast<const float2*>(d_in);
    a.out = static_cast<float2*>(d_out);
    a.hist = reinterpret_cast<const float2*>(e->d_hist[e->cur]);
    a.hist_keep = a.hist;
    a.hist_next = reinterpret_cast<float2*>(e->d_hist[e->cur ^ 1]);
    rc = raw_history(e, s, &a.hist, &a.hist_raw_next);
    if (rc) return rc;
    a.tables = e->d_pfb;
    a.count = count;
    a.nout = nout;
    a.H = e->H;
    a.PH = pfb_phases(e);
    a.Q = pfb_Q(e);
    a.Lo = qk::kPfbF + 1 - a.Q;
    // column outputs n' the call needs: output n = PH n' - phi, n < nout  ->  n' <= ceil((nout - 1) / PH)
    const int64_t ncol = a.PH == 1 ? nout : (nout + a.PH - 2) / a.PH + 1;
    a.nseg = (int)((ncol + a.Lo - 1) / a.Lo);
    // persistent workgroups of 4 waves, one segment per wave at a time; 2 workgroups resident per CU (72 KB of LDS,
    // ~220 VGPRs), QDSP_HIP_PFB_WG_PER_CU queued per CU
    // (decimate by 4: 8 waves per workgroup -- they share the two spectrum tables, 159 KB of LDS -- one workgroup per CU)
    a.real = e->ch == 1 ? 1 : 0;
    const int wpw = 4 * a.PH;
    int nwg = 256 * qk::knob(qk::K_PFB_WG_PER_CU, 2) / a.PH;
    if (nwg > 1024) nwg = 1024;
    const int nunits = a.real ? (a.nseg + 1) / 2 : a.nseg;      // (real data: a wave takes a PAIR of segments at a time)
    const int need = (nunits + wpw - 1) / wpw;
    if (nwg > need) nwg = need;
    if (nwg < 1) nwg = 1;
    a.nwg = nwg;
    a.rot = e->rotate ? 1 : 0;
    if (a.rot) {
        a.phase_in0 = e->phase;
        a.phase0 = e->phase - (unsigned long long)(e->ntaps - 1) * e->dphase;
        a.dphase = e->dphase;
        a.gm1 = e->volk_gain ? e->gm1 : 0.0f;
        unit_of_fx_c(e->dphase, 8.0L * (long double)a.Lo * (long double)(wpw * nwg), &a.rot_step.x, &a.rot_step.y);
        if (wpw * nwg > 4096) return QDSP_HIP_EINVAL;     // (seg_pow covers wave indices below 2^12)
        unit_of_fx(a.phase0 - (unsigned long long)(8 * (a.Q - 1) + 1) * a.dphase, 1.0L, &a.pb_base.x, &a.pb_base.y);
        for (int k = 0; k < 12; k++) unit_of_fx_c(e->dphase, 8.0L * (long double)a.Lo * (long double)(1 << k), &a.seg_pow[k].x, &a.seg_pow[k].y);
        for (int b1 = 0; b1 < 8; b1++) {
            double c, sn;
            unit_of_fx_c(e->dphase, (long double)(512 * b1), &c, &sn);
            a.wtab[b1] = make_float2((float)c, (float)sn);
        }
    }
    rc = qk::launch_pfb_dec(a, s);
    if (rc) return rc;
    e->raw_valid = e->rotate && e->H > 0;   // (the caller flips cur: the raw hand-over then sits at d_hist_raw[cur])
    e->last.name = a.real ? (a.PH == 2 ? "pfb_dec4_real_kernel" : "pfb_dec8_real_kernel") : a.PH == 2 ? "pfb_dec4_kernel" : "pfb_dec8_kernel";
    e->last.grid = nwg + 1;
    e->last.block = qk::kPfbNT * a.PH;
    e->last.lds = (int)((qk::kPfbTableElems + (a.PH - 1) * qk::kPfbG1Elems + wpw * 64 * qk::kPfbRow + (a.PH - 1) * wpw * 64 * 9) * sizeof(float2));
    return 0;
}

int launch_fft(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s) {
    if (pfb_eligible(e, count)) return launch_pfb(e, d_in, count, nout, d_out, s);
    if (fft1k_eligible(e, count)) return launch_fft1k(e, d_in, count, nout, d_out, s);
    int rc = fft_prepare(e);
    if (rc) return rc;
    qk::FftArgs a;
    memset(&a, 0, sizeof(a));
    a.in = static_cast<const float2*>(d_in);
    a.out = static_cast<float2*>(d_out);
    a.hist = reinterpret_cast<const float2*>(e->d_hist[e->cur]);
    a.hist_keep = a.hist;
    a.hist_next = reinterpret_cast<float2*>(e->d_hist[e->cur ^ 1]);
    rc = raw_history(e, s, &a.hist, &a.hist_raw_next);
    if (rc) return rc;
    a.Hf = e->d_fft_H;
    a.TA = e->d_fft_TA;
    a.TB = e->d_fft_TB;
    a.count = count;
    a.nout = nout;
    a.H = e->H;
    a.dec = fft_dec(e);
    // Decimate by 2 on chip-filling calls (round 3, profiles/r03_tune_dec2.txt): the pruned inverse saves one radix-16 pass on half the
    // lanes but runs its last two passes on 128 of them behind two more barriers and stores 8 bytes per lane; the full inverse with every
    // other output kept is ahead from 2^25 samples (0.115 against 0.128 ms; 0.400 against 0.491 at 2^27), with the NCO from 2^26
    // (0.247 / 0.254; 0.472 / 0.505).  Decimations 4, 8, 16 keep the pruned form at every size (0.33 / 0.30 / 0.29 against 0.38 / 0.37 / 0.39).
    if (a.dec == 2 && e->ch == 2) {
        const int lim = qk::knob(qk::K_FFT_PRUNE2_MAX_COUNT, -1);
        // (re-measured once both forms ran four workgroups per CU where they can: 2^24 samples 0.0606 against 0.0623 ms, NCO from 2^25: 0.126 / 0.131)
        if (count >= (lim >= 0 ? (int64_t)lim : (int64_t)(e->rotate ? 1 << 25 : 1 << 24))) a.dec = 1;
    }
    a.rot = e->rotate ? 1 : 0;
    a.decm = 1;
    if (a.dec == 1 && e->kind != KIND_FIR) {
        // any-decimation resampler / VFO: y[n'] sits at stream position n'*M - 1.  Even overlap and
        // an even segment start (two samples before the first valid position) keep 16-byte loads.
        a.decm = e->M;
        a.decm_inv = (e->M >= 2 && (unsigned long long)(e->M + 4096) * (unsigned long long)e->M < (1ULL << 32)) ? (unsigned)((1ULL << 32) / (unsigned)e->M) + 1u : 0u;
        a.strided = 1;
        a.ov = (e->ntaps - 1 + 1) & ~1;
        a.seg_shift = a.ov + 2;
        a.L = qk::kFftN - a.ov;
        a.nblocks = (int)((count + 2 + a.L - 1) / a.L);
        a.vec = (((uintptr_t)d_in) & 15) == 0 && !0;
    } else if (a.dec == 1) {
        a.ov = (e->ntaps - 1 + 1) & ~1;   // FIR: out index == stream position; even so segments stay 16-byte aligned
        a.vec = ((((uintptr_t)d_in) | ((uintptr_t)d_out)) & 15) == 0 && !0;
        a.seg_shift = a.ov;
        a.L = qk::kFftN - a.ov;
        a.nblocks = (int)((count + a.L - 1) / a.L);
    } else {
        // resampler: y[n'] sits at stream position n'*dec - 1; segments start one sample early
        a.ov = ((e->ntaps - 1 + a.dec - 1) / a.dec) * a.dec;
        a.seg_shift = a.ov + 1;
        a.L = qk::kFftN - a.ov;
        const int per_block = a.L / a.dec;
        a.nblocks = (int)((nout + per_block - 1) / per_block);
    }
    if (e->ch == 1) {   // real data: one workgroup iteration = a PAIR of real segments
        a.real2 = 1;
        // 8-byte pair accesses: segments start on even samples (L, ov, seg_shift are even) of 8-byte aligned buffers
        a.vec = ((((uintptr_t)d_in) | ((uintptr_t)d_out)) & 7) == 0 && !0;
        a.nblocks = (a.nblocks + 1) / 2;
    }
    // FIR: 4 workgroups resident per CU (124 VGPRs, 37 KB LDS), 16 queued per CU for balance.
    // Decimators work in groups of `dec` segments, 2 resident per CU (70 KB LDS).
    // (grouped only when the groups fill the chip -- measured crossover ~1000 groups, 2^25 samples at decimation 8 --
    // below that the segments run one per workgroup: a 1e6-sample block 10 us instead of 28)
    // FIR<complex_t> on aligned buffers: the LDS-DMA form (fir_fft_dma_kernel, fft_fir.hip)
    a.nt = qk::knob(qk::K_FFT_NT, 0);
    a.abl = qk::knob(qk::K_FFT_ABL, 0);
    a.stamps = reinterpret_cast<unsigned*>(qk::knobs_snapshot()->fft_stamps);   // diagnostic: scripts/stamp_fir_fft.py
    a.dma = (a.dec == 1 && !a.strided && !a.rot && e->ch == 2 && a.vec && a.ov <= 2048) ? qk::knob(qk::K_FFT_DMA, 1) : 0;   // 1: scalar arithmetic, 2: packed
    const bool grouped = a.dec >= 4 && (a.nblocks + a.dec - 1) / a.dec >= 1024;
    a.grouped = grouped ? 1 : 0;
    const int per_cu = qk::knob(qk::K_FFT_WG_PER_CU, grouped ? 4 : 16);
    const int units = grouped ? (a.nblocks + a.dec - 1) / a.dec : a.nblocks;
    int nwg = 256 * per_cu;
    if (nwg > units) nwg = units;
    a.nwg = nwg;
    if (a.rot) {
        a.phase_in0 = e->phase;
        a.phase0 = e->phase - (unsigned long long)(e->ntaps - 1) * e->dphase;
        a.dphase = e->dphase;
        a.gm1 = e->volk_gain ? e->gm1 : 0.0f;
        // one workgroup's step between its units: nwg segments (per-segment kernel), nwg groups of dec segments (grouped)
        const long double step_mult = (long double)nwg * (long double)a.L * (long double)(grouped ? a.dec : 1);
        if (!e->rot_step_ok || e->rot_step_dphase != e->dphase || e->rot_step_mult != step_mult) {
            unit_of_fx_c(e->dphase, step_mult, &e->rot_step_val.x, &e->rot_step_val.y);
            e->rot_step_dphase = e->dphase;
            e->rot_step_mult = step_mult;
            e->rot_step_ok = true;
        }
        a.rot_step = e->rot_step_val;
        if (!e->wtab4k_ok || e->wtab4k_dphase != e->dphase) {
            for (int n2 = 0; n2 < 16; n2++) {
                double c, sn;
                unit_of_fx_c(e->dphase, (long double)(256 * n2), &c, &sn);
                e->wtab4k[n2] = make_float2((float)c, (float)sn);
            }
            e->wtab4k_dphase = e->dphase;
            e->wtab4k_ok = true;
        }
        memcpy(a.wtab, e->wtab4k, sizeof(a.wtab));
    }
    rc = qk::launch_fir_fft(a, nwg + 1, s);
    if (rc) return rc;
    e->raw_valid = e->rotate && e->H > 0;   // (the caller flips cur: the raw hand-over then sits at d_hist_raw[cur])
    e->last.name = a.dma == 2 ? "fir_fft_dmapk_kernel" : a.dma ? "fir_fft_dma_kernel" : "fir_fft_kernel";
    e->last.grid = nwg + 1;
    e->last.block = qk::kFftNT;
    e->last.lds = (int)(((grouped ? 2 : 1) * qk::kFftLdsElems + 16 * 17) * sizeof(float2));
    return 0;
}

int launch_xlate_inc(Engine* e, const void* d_in, int64_t count, void* d_out, unsigned long long phase0, unsigned long long dphase, float gm1,
                     hipStream_t s);
int launch_xlate_raw(Engine* e, const void* d_in, int64_t count, void* d_out, unsigned long long phase0, float gm1, hipStream_t s) {
    return launch_xlate_inc(e, d_in, count, d_out, phase0, e->dphase, gm1, s);
}

int launch_xlate(Engine* e, const void* d_in, int64_t count, void* d_out, hipStream_t s) {
    return launch_xlate_raw(e, d_in, count, d_out, e->phase, e->volk_gain ? e->gm1 : 0.0f, s);
}

int launch_xlate_inc(Engine* e, const void* d_in, int64_t count, void* d_out, unsigned long long phase0, unsigned long long dphase, float gm1,
                     hipStream_t s) {
    constexpr int NT = 256;
    if (count <= 0) return 0;
    qk::XlateArgs a;
    a.in = static_cast<const float2*>(d_in);
    a.out = static_cast<float2*>(d_out);
    a.count = count;
    a.phase0 = phase0;
    a.dphase = dphase;
    const long long npairs = (count + 1) / 2;
    long long grid = (npairs + NT - 1) / NT;
    // one pair per lane up to 2^27 samples: measured 0.345 ms per 2^27 samples against 0.46 ms with 16 blocks per CU
    // looping 64 times (the per-lane FP64 sincos is cheaper than the lost memory-level parallelism)
    { const long long cap = 256LL * 1024; if (grid > cap) grid = cap; }
    unit_of_fx_c(dphase, 1.0L, &a.rot_one.x, &a.rot_one.y);
    unit_of_fx_c(dphase, (long double)(2 * grid * NT), &a.rot_stride.x, &a.rot_stride.y);
    a.vec = (((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0;   // d_in == nullptr (SineSource) counts as aligned
    a.gm1 = gm1;
    hipLaunchKernelGGL((qk::xlate_kernel<NT>), dim3((unsigned)grid), dim3(NT), 0, s, a);
    HIPCHK(hipGetLastError());
    e->last.name = "xlate_kernel";
    e->last.grid = (int)grid;
    e->last.block = NT;
    e->last.lds = 0;
    return 0;
}

// outputs per wave task: a task reads one tile of 16 rows beyond its own (T = 256: 6 %); small calls take shorter
// tasks so that a reference-sized block still spreads over the chip (1e6 samples at decimation 50: 20 000 outputs)
void mf_tasks(qk::MfArgs& a, int64_t nout, int nchan, bool rot, bool real) {
    long long T = nout * nchan / 2048;      // (one round of the chip: 3072 wave slots; 8192 left a 4M-sample call at T = 16 -- half of every task's reads its neighbour's -- and 18 us instead of 14)
    T = (T + 15) / 16 * 16;
    // (real data, round 4: 256 outputs per task on chip-filling calls -- decimate-by-50 0.118 -> 0.110 ms per 2^27 samples, by-200 -10 %, by-32 +1 %;
    // at 2^23 samples 128 keeps twice the tasks and decimate-by-16 a third quicker)
    const long long tmax = qk::knob(qk::K_MF_TASK_MAX, (rot || (real && a.count >= (1LL << 26))) ? 256 : 128);
    if (T > tmax) T = tmax;
    if (T < 16) T = 16;
    a.T = (int)T;
    a.ntasks = (int)((nout + T - 1) / T);
    a.minv = (unsigned)(((1ull << 32) + a.M - 1) / a.M);
    a.defer_store = (long long)a.count * nchan >= (1LL << 22);      // (profiles/r04_mf_deferred_store.txt)
}
// NCO tables: one tile (16 M samples) further in FP64, load i of a tile (64 i samples) in FP32
void mf_rot_tables(unsigned long long dphase, int M, int KJ, double2* step, float2* rot_k) {
    unit_of_fx_c(dphase, 16.0L * (long double)M, &step->x, &step->y);
    for (int k = 0; k < 2 * KJ; k++) {
        double c, sn;
        unit_of_fx_c(dphase, 64.0L * (long double)k, &c, &sn);
        rot_k[k] = make_float2((float)c, (float)sn);
    }
}

// FIR<complex_t> on reference-sized calls: the direct form arranged for latency (fir_lat.hip.h), bit-identical to
// fir_core_kernel.  Per call at 256 taps the overlap-save kernel takes 9.0-9.3 us at 16 384-262 144 samples and
// fir_core_kernel 11.3+ (profiles/r02_tune_call_size.txt).
bool fir_lat_eligible(const Engine* e, int64_t count) {
    if (e->kind != KIND_FIR || e->ch != 2 || !e->has_filter || e->L != 1 || e->M != 1) return false;
    if (e->ntaps > 1024 || qk::knob(qk::K_NO_FIR_LAT, 0)) return false;
    if (e->auto_pick) return e->auto_pick == PICK_LAT && count > 0;      // the measured table has spoken (dispatch_table.inc)
    // (a wave walks all taps of its 64 outputs: 600 taps take 9.3 us on 4096 samples, the one-wave overlap-save kernel 5.6)
    if (e->ntaps > 320 && fft1k_eligible(e, count)) return false;
    // measured: 2.5 us + 1.7e-7 us per tap and sample (63 / 127 / 256 taps at 65 536 samples: 2.9 / 3.6 / 5.3 us); what it
    // competes with is the overlap-save kernel (~9 us up to 262 144 samples) from 96 taps on, fir_core_kernel (5.7-6.1 us) below
    // -- and, for up to 769 taps, the one-wave-per-segment overlap-save kernel (5.2-5.9 us up to 262 144 samples whatever the taps)
    const int64_t limit = qk::knob(qk::K_FIR_LAT_MAX_WORK, e->ntaps > 769 ? 1 << 25 : 1 << 24);
    return count > 0 && count * (int64_t)e->ntaps <= limit;
}
int launch_fir_lat(Engine* e, const void* d_in, int64_t count, void* d_out, hipStream_t s) {
    qk::FirLatArgs a;
    memset(&a, 0, sizeof(a));
    a.in = static_cast<const float2*>(d_in);
    a.out = static_cast<float2*>(d_out);
    a.hist = reinterpret_cast<const float2*>(e->d_hist[e->cur]);
    a.hist_next = reinterpret_cast<float2*>(e->d_hist[e->cur ^ 1]);
    a.taps = e->d_taps;
    a.count = count;
    a.N = e->ntaps;
    a.Np = (e->ntaps + 7) & ~7;
    a.nwaves = (int)((count + 63) / 64);
    const int rc = qk::launch_fir_lat(a, s);
    if (rc) return rc;
    e->last.name = "fir_lat_kernel";
    e->last.grid = (a.nwaves + 3) / 4 + 1;
    e->last.block = 256;
    e->last.lds = (int)qk::fir_lat_lds_bytes(a.Np);
    return 0;
}

int launch_rm(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s) {
    qk::RmArgs a;
    memset(&a, 0, sizeof(a));
    a.in = static_cast<const float2*>(d_in);
    a.out = static_cast<float2*>(d_out);
    a.hist = reinterpret_cast<const float2*>(e->d_hist[e->cur]);
    a.hist_next = reinterpret_cast<float2*>(e->d_hist[e->cur ^ 1]);
    a.atab = e->d_taps_rm;
    a.count = count;
    a.nout = nout;
    a.L = e->rm_J * e->L;          // (J merged periods)
    a.M = e->rm_J * e->M;
    a.P = e->P;
    a.minv = (unsigned)(((1ull << 32) + a.M - 1) / a.M);
    a.qpb = e->rm_qpb;
    a.ngrp = e->rm_ngrp;
    a.KB = e->rm_KB;
    a.ext = e->rm_ext;
    a.pitch = e->rm_pitch;
    a.G = e->rm_G;
    a.total = 4 * a.G * a.M + a.ext;
    const long long nper = (nout + a.L - 1) / a.L;
    a.ntiles = (int)((nper + 4 * a.G - 1) / (4 * a.G));
    // waves QUEUED per SIMD (three are resident): more, shorter-lived waves than fit even out the tail and keep the tiles in flight closer
    // together in memory (round 3, scripts/tune_rm_waves.py, profiles/r03_tune_rm_waves.txt: 147/160 0.236 -> 0.224 ms per 2^26 samples,
    // interpolate-by-6 0.164 -> 0.130, 10/7 0.181 -> 0.179; the small decimating ratios are indifferent)
    int nwaves = 1024 * qk::knob(qk::K_RM_WAVES_PER_SIMD, e->M == 1 ? 12 : e->L >= 100 ? 8 : 6);
    if (nwaves > a.ntiles) nwaves = a.ntiles;
    if (nwaves < 1) nwaves = 1;
    a.nwaves = nwaves;
    if (e->rotate) {
        a.phase0 = e->phase;
        a.dphase = e->dphase;
        a.gm1 = e->volk_gain ? e->gm1 : 0.0f;
        unit_of_fx_c(e->dphase, (long double)nwaves * 4.0L * (long double)a.G * (long double)a.M, &a.rot_step.x, &a.rot_step.y);
        for (int k = 0; k < qk::kRmNE; k++) {
            double c, sn;
            unit_of_fx_c(e->dphase, 64.0L * (long double)k, &c, &sn);
            a.rot_k[k] = make_float2((float)c, (float)sn);
        }
    }
    const int rc = qk::launch_rm_resamp(a, e->rotate, e->ch == 1, s);
    if (rc) return rc < 0 && rc != -1 ? rc : QDSP_HIP_EINVAL;
    e->last.name = e->ch == 1 ? "resamp_mfma_real_kernel" : "resamp_mfma_kernel";
    e->last.grid = (nwaves + 3) / 4 + 1;
    e->last.block = 256;
    e->last.lds = (int)qk::rm_lds_bytes(a.ngrp, a.KB, a.G, a.pitch, e->ch == 1);
    return 0;
}

int launch_mf(Engine* e, const void* d_in, int64_t count, int64_t nout, void* d_out, hipStream_t s) {
    qk::MfArgs a;
    memset(&a, 0, sizeof(a));
    a.in = static_cast<const float2*>(d_in);
    a.out = static_cast<float2*>(d_out);
    a.hist = reinterpret_cast<const float2*>(e->d_hist[e->cur]);
    a.hist_next = reinterpret_cast<float2*>(e->d_hist[e->cur ^ 1]);
    a.tapk = e->d_taps_mf;
    a.count = count;
    a.nout = nout;
    a.P = e->P;
    a.M = e->mf_keep2 ? e->M / 2 : e->M;
    a.keep2 = e->mf_keep2;
    if (e->mf_keep2) a.nout = 2 * nout - 1;      // (in the kernel's units: outputs of the decimator by M / 2)
    // the kernel's unguarded loads of an interior tile read 64 (2 KJ - 2) samples from the tile's start: they must lie inside the tile's own 16 M
    // (true for KJ = ceil(M / 8); a plan that pads KJ beyond that would read past the end of the input -- found the hard way, round 4)
    if (128 * (e->mf_KJ - 1) > 16 * a.M) return QDSP_HIP_EINVAL;
    mf_tasks(a, a.nout, 1, e->rotate, e->ch == 1);
    if (e->rotate) {
        a.phase0 = e->phase;
        a.dphase = e->dphase;
        a.gm1 = e->volk_gain ? e->gm1 : 0.0f;
        mf_rot_tables(e->dphase, a.M, e->mf_KJ, &a.rot_step, a.rot_k);
    }
    const int rc = qk::launch_mf_dec(a, e->mf_KJ, e->rotate, qk::knob(qk::K_MF_DEPTH, (e->mf_KJ <= 8 && !(e->rotate && e->mf_KJ >= 5)) ? 2 : 1)      /* (fused VFO from decimation 33 on: one tile ahead = 118 VGPRs = four waves per SIMD, 1-3 % ahead of two tiles at three: 0.232 -> 0.229 ms at decimation 50; the plain decimator the other way round) */, e->mf_QS, e->ch == 1, s);
    if (rc) return rc < 0 && rc != -1 ? rc : QDSP_HIP_EINVAL;
    e->last.name = e->ch == 1 ? "decim_mfma_real_kernel" : "decim_mfma_kernel";
    e->last.grid = (a.ntasks + 3) / 4 + 1;
    e->last.block = 256;
    e->last.lds = e->ch == 1 ? 4 * (16 * (8 * e->mf_KJ + 4) + 64) * (int)sizeof(float) : 4 * (16 * (8 * e->mf_KJ + 2) + 64) * (int)sizeof(float2);
    return 0;
}

// Smallest call the MFMA kernels take (profiles/r02_tune_call_size_mfma.txt).  Their waves set up operand tables, slot
// maps and an FP64 phasor before the first tile and then walk their tiles one after the other, so on a reference-sized
// block (<= 1e6 samples, src/dsp/stream.h:7) the general direct kernel is 1-3 us quicker wherever a wave's tile is large
// (decimations past 32, periods of >= 33 outputs); the crossover sits at 2-5e6 samples -- 1.4e7 for the decimations
// run at half the row length -- and from there on the MFMA kernels are 1.3-2x faster.  Short rows (decimation <= 32)
// and the small rational ratios are quicker at every size.
// Reference-sized calls of the small-interpolation resamplers: resamp_lm_kernel gives every lane R x L accumulators over all
// P taps -- the throughput form (0.27-0.30 ms per 2^26 samples where the general kernel needs 0.43-0.60) -- so a call
// too small to fill the chip takes as long as ONE lane's chain: 6.5 us at 32 taps per phase, 14 us at 67 (3/7, 200
// taps), 24-28 us at 128 (5/8, 640 taps) or on 10/7, whatever the size up to ~1e6 samples.  The general kernel with
// its call-sized tiles (any_plan) takes 4.0-4.7 us there and grows with outputs x taps.  Measured crossovers
// (profiles/r02_tune_small_resamp.txt), in outputs of the call: 16 384 x taps per phase at decimation 3-4, 2x that from
// decimation 5, 2.5x at decimation 1-2, never beyond 2^20 outputs; filters under 16 taps per phase stay (3.8-4.5 us).
// (interp 10, decim 7 past the MFMA kernel's tap range: the general kernel at every size -- 0.58 against 0.74 ms per 2^26)
bool lm_yields_to_any(const Engine* e, int64_t nout) {
    if (qk::knob(qk::K_NO_LM_SMALL_CALL_RULE, 0)) return false;
    if (e->L == 10 && e->M == 7) return true;
    if (e->P < 16) return false;
    int64_t lim = (int64_t)e->P * (e->M >= 5 ? 32768 : e->M >= 3 ? 16384 : 40960);
    if (lim > (1 << 20)) lim = 1 << 20;
    return nout <= lim;
}

// The strided-window decimator on reference-sized calls: 6.0-8.2 us per call at 95-127 taps (decimation 2 / 4 / 8) where
// the one-wave overlap-save kernel takes 5.5 whatever the taps; from ~1e6 samples on the window kernel is ahead again
// (profiles/r02_tune_fft1k.txt).
bool win_yields_to_fft1k(const Engine* e, int64_t count) {
    if (e->ch != 2 || e->ntaps < 96 || count > (1 << 19) || 0) return false;   // (real data: 5.6-6.8 us against 7.1-8.3)
    return fft1k_eligible(e, count);
}

// The strided-window decimator on chip-filling calls (round 3, scripts/tune_dec_small.py, profiles/r03_tune_win_vs_fft.txt, 2^27 samples):
// its time grows with the taps (decimation 2: 0.32 ms at 48 taps, 0.53 at 128) while the overlap-save forms cost the same at any
// length -- 0.48 / 0.38 / 0.33 ms at decimation 2 / 3 / 4, pfb_dec8_kernel 0.24-0.25 at decimation 8 -- so past the measured
// crossovers the big calls go there (decimation 8, 160 taps: 0.30 -> 0.25 ms, fused VFO 0.35 -> 0.25; decimation 2, 144 taps:
// 0.57 -> 0.48).  use_win()'s limits, measured on 2^26-sample calls in round 1, keep the smaller calls.
bool win_yields_to_fft_big(const Engine* e, int64_t count) {
    if (e->ch != 2 || 0) return false;
    const int M = e->M, P = e->P;
    if (M == 8) return pfb_eligible(e, count) && P >= (e->rotate ? 56 : 88);
    // (at 2^24 samples the window kernel is still ahead at decimation 3-5 -- 0.046-0.049 against 0.053-0.055 ms at 112-160 taps --
    // and level at decimation 2)
    if (M == 2) return count >= (1 << 24) && P >= 104;
    if (M == 3 || M == 4) return count >= (1 << 26) && P >= 104;
    if (M == 5) return count >= (1 << 26) && !e->rotate && P >= 152;
    if (M == 6) return count >= (1 << 26) && !e->rotate && P >= 208;      // (256 taps: 0.455 against 0.37 ms; level at 192)
    return false;
}

// Real data, chip-filling calls (same sweep): the de-interleaved direct kernel moves 4-byte samples at decimation 2 / 4 faster than the
// strided window does (decimation 2: 0.14-0.25 against 0.18-0.36 ms at 32-128 taps; decimation 4: 0.12-0.18 against 0.15-0.23), and from
// ~100 taps at decimation 5 / 8.
bool real_win_yields_to_core_big(const Engine* e, int64_t count) {
    if (e->ch != 1 || count < (1 << 22) || !use_core(e) || 0) return false;
    return e->M == 2 || e->M == 4 || (e->M == 5 && e->P >= 96) || (e->M == 8 && e->P >= 128);
}

int64_t mf_min_count(const Engine* e) {
    const int v = qk::knob(qk::K_MF_MIN_COUNT, -1);
    if (v >= 0) return v;
    if (e->mf_keep2) return 1 << 24;
    // (filters of at most three taps per unit of decimation: two or three tap rows of the operand in use -- the general kernel is 10-25 % ahead at
    // 2^22 samples, level at 2^24; profiles/r03_sweep_mid.txt)
    if (e->P <= 3 * e->M) return 8 << 20;
    if (e->mf_QS == 2 || e->mf_KJ > 4) return 3 << 20;
    return 0;
}
int64_t rm_min_count(const Engine* e) {
    const int v = qk::knob(qk::K_RM_MIN_COUNT, -1);
    if (v >= 0) return v;
    // (profiles/r03_sweep_rm_grid.txt, second part: decimating ratios are ahead from 2^22 input samples; interpolating ones -- whose work follows
    // the OUTPUT count -- only from ~10 million inputs: 8/3 at 6.3 million x 1.2, at 12.6 million x 0.9)
    int64_t base = e->rm_big_only ? (e->L > e->M ? 12 << 20 : 1 << 22) : (e->L >= 33 ? 6 << 20 : 0);
    if (e->ch == 1) {
        // real data (profiles/r04_real_rational.txt): interpolating ratios only pay from 2^25 samples on (6/1 at 2^20: 2-3.3x slower than the general
        // kernel, at 2^23 1.07-1.23x, at 2^26 0.53-0.74x; 48/5 at 2^23 1.09-1.33x; 160/147 level at 2^23); the decimating ones at every size up
        // to 19 taps per phase, from 2^22 samples beyond that (3/8 with 20 taps per phase at 2^20: 1.31x)
        if (e->L > e->M && base < (1 << 25)) base = 1 << 25;
        if (e->L < e->M && e->P >= 20 && base < (1 << 22)) base = 1 << 22;
    }
    return base;
}

// One run() worth of work on device pointers.  Returns the output count.
int64_t process_dev(Engine* e, const void* d_in, int64_t count, void* d_out, void* stream) {
    if (count < 0 || (count > 0 && ((!d_in && e->kind != KIND_SINE) || !d_out))) return QDSP_HIP_EINVAL;
    apply_pending_inc(e);
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);  // NULL == HIP's default stream
    const int64_t nout = out_size(e, count);
    int rc = 0;
    bool took_fft = false;
    e->auto_pick = PICK_NONE;
    e->auto_veto = 0;
    e->auto_mode = 0;
    const bool cplx_dec = (e->kind == KIND_DECIM || e->kind == KIND_VFO) && e->ch == 2 && e->M >= 2;
    const bool real_any = (e->kind == KIND_DECIM || e->kind == KIND_FIR) && e->ch == 1 && !e->rotate;
    if ((cplx_dec || real_any) && e->has_filter && e->L == 1 && e->fir_mode == 0 && qk::knob(qk::K_FIR_MODE, 0) == 0) {
        // Integer decimators and the fused VFO on complex data, FIR / decimators on real data: the rule chain below decides, EXCEPT where the measured table
        // (decim_table.inc <- profiles/r04_sweep_decim_table.txt, scripts/gen_dispatch_table.py) found one of eight switch settings more than
        // 4 % faster in the call's cell.  QDSP_HIP_DECIM_SETTING = 1..8 forces a setting (the sweep, the regression test), 0 = rules only.
        const int forced = qk::knob(qk::K_DECIM_SETTING, -1);
        const int setting = forced >= 0 ? forced : decim_table_setting(real_any ? 2 : e->rotate ? 1 : 0, e->M, e->ntaps, count);
        e->auto_veto = kDecimSettingVeto[setting & 15];
        e->auto_mode = kDecimSettingMode[setting & 15];
    }
    if (e->kind == KIND_FIR && e->ch == 2 && e->has_filter && mode_of(e) == 0) {
        const int forced = qk::knob(qk::K_FIR_PICK, 0);               // 1..4: the sweep and the regression test force a family
        e->auto_pick = forced >= PICK_LAT && forced <= PICK_FFT4K ? forced : qk::knob(qk::K_NO_FIR_TABLE, 0) ? PICK_NONE : fir_table_pick(count, e->ntaps);
        // (QDSP_HIP_FFT1K_MAX_COUNT, the tests' way of pinning the 4096-point kernels at oracle-sized inputs, outranks the table)
        if (e->auto_pick == PICK_FFT1K && !forced) {
            const int cap = qk::knob(qk::K_FFT1K_MAX_COUNT, -1);
            if ((cap >= 0 && count > cap) || qk::knob(qk::K_NO_FFT1K, 0)) e->auto_pick = PICK_FFT4K;
        }
        // structural limits of the family named: none for the direct form; the others fall back to the rule chain
        if ((e->auto_pick == PICK_LAT && e->ntaps > 1024) || (e->auto_pick == PICK_FFT1K && e->ntaps > 769) ||
            ((e->auto_pick == PICK_FFT1K || e->auto_pick == PICK_FFT4K) && !fft_dec(e)))
            e->auto_pick = PICK_NONE;
    }
    if (!e->has_filter) {
        rc = launch_xlate(e, d_in, count, d_out, s);
    } else if (mode_of(e) == 0 && fir_lat_eligible(e, count)) {
        rc = launch_fir_lat(e, d_in, count, d_out, s);
        if (rc == 0) e->cur ^= 1;
    } else if (e->d_taps_mf && mode_of(e) == 0 && nout > 0 && count >= mf_min_count(e) && !qk::knob(qk::K_NO_MF, 0) && !(e->auto_veto & VETO_MF)) {
        // large integer decimations (the VFO's usual job) as an FP32 matrix product on the MFMA units (mf_dec.hip.h)
        rc = launch_mf(e, d_in, count, nout, d_out, s);
        if (rc == 0) e->cur ^= 1;
    } else if (fft_eligible(e, count) && !(mode_of(e) == 0 && use_win(e) && e->d_taps_lm && !win_yields_to_fft1k(e, count) && !win_yields_to_fft_big(e, count))) {
        rc = launch_fft(e, d_in, count, nout, d_out, s);
        if (rc == 0) e->cur ^= 1;
        took_fft = true;
    } else if (use_win(e) && e->d_taps_lm && mode_of(e) == 0 && !real_win_yields_to_core_big(e, count)) {
        // AUTO only: QDSP_HIP_FIR_DIRECT keeps meaning fir_core_kernel (the form the bit-exactness tests pin),
        // QDSP_HIP_FIR_FFT the overlap-save kernels
        if (e->ch == 2) rc = e->rotate ? launch_win<2, true>(e, d_in, count, nout, d_out, s) : launch_win<2, false>(e, d_in, count, nout, d_out, s);
        else rc = launch_win<1, false>(e, d_in, count, nout, d_out, s);
        if (rc == 0) e->cur ^= 1;
    } else if (use_core(e)) {
        qk::CoreArgs a;
        memset(&a, 0, sizeof(a));
        a.in = d_in;
        a.out = d_out;
        a.hist = e->d_hist[e->cur];
        a.hist_next = e->d_hist[e->cur ^ 1];
        a.taps = e->d_taps;
        a.count = count;
        a.nout = nout;
        a.H = e->H;
        a.M = e->M;
        a.Q = (e->P + e->M - 1) / e->M;
        a.phase0 = e->phase;
        a.dphase = e->dphase;
        a.gm1 = e->volk_gain ? e->gm1 : 0.0f;
        if (e->ch == 2) rc = e->rotate ? launch_core<2, true>(e, a, s) : launch_core<2, false>(e, a, s);
        else rc = launch_core<1, false>(e, a, s);
        if (rc == 0) e->cur ^= 1;
    } else if (e->d_taps_rm && e->rm_ngrp && mode_of(e) == 0 && nout > 0 && count >= rm_min_count(e) && !qk::knob(qk::K_NO_RM, 0)) {
        // rational ratios with interp >= 6 (48 kHz <-> 44.1 kHz is 147 / 160) on the MFMA units (rm_resamp.hip.h)
        rc = launch_rm(e, d_in, count, nout, d_out, s);
        if (rc == 0) e->cur ^= 1;
    } else if (use_lm(e) && e->d_taps_lm && !(mode_of(e) == 0 && lm_yields_to_any(e, nout))) {
        if (e->ch == 2) rc = e->rotate ? launch_lm<2, true>(e, d_in, count, nout, d_out, s) : launch_lm<2, false>(e, d_in, count, nout, d_out, s);
        else rc = launch_lm<1, false>(e, d_in, count, nout, d_out, s);
        if (rc == 0) e->cur ^= 1;
    } else {
        qk::AnyArgs a;
        memset(&a, 0, sizeof(a));
        a.in = d_in;
        a.out = d_out;
        a.hist = e->d_hist[e->cur];
        a.hist_next = e->d_hist[e->cur ^ 1];
        a.phases = e->d_taps;
        a.count = count;
        a.nout = nout;
        a.L = e->L;
        a.M = e->M;
        a.P = e->P;
        a.phase0 = e->phase;
        a.dphase = e->dphase;
        a.gm1 = e->volk_gain ? e->gm1 : 0.0f;
        if (e->ch == 2) rc = e->rotate ? launch_any<2, true>(e, a, s) : launch_any<2, false>(e, a, s);
        else rc = launch_any<1, false>(e, a, s);
        if (rc == 0) e->cur ^= 1;
    }
    if (!took_fft) e->raw_valid = false;   // (the direct kernels hand over the rotated history only)
    if (rc) return rc;
    if (e->rotate) e->phase += (unsigned long long)count * e->dphase;  // exact mod 2^64
    return nout;
}

// Host-pointer path: what a block's run() calls between _in->read() and out.swap().
int64_t process_host(Engine* e, const float* in, int count, float* out) {
    if (count < 0 || (count > 0 && (!in || !out))) return QDSP_HIP_EINVAL;
    if (count > e->max_block) {
        int rc = ensure_io(e, count);
        if (rc) return rc;
    }
    HIPCHK(hipSetDevice(e->device));
    const size_t in_bytes = (size_t)count * e->ch * sizeof(float);
    // (Round 4 re-measured pieces of one synchronous call on two streams -- upload stream + this one: 0.321 against 0.315 ms for a 1e6-sample FIR
    // block, 0.261 against 0.194 for the decimator: inside ONE call the two copy directions still do not overlap.  Across the calls of a block
    // graph they do: process_ex, Engine::up_stream.)
    if (e->up_stream && e->kernel_recorded) HIPCHK(hipStreamWaitEvent(e->stream, e->ev_up, 0));   // (an earlier split upload of this handle)
    if (count) HIPCHK(hipMemcpyAsync(e->d_in, in, in_bytes, hipMemcpyHostToDevice, e->stream));
    const int64_t nout = process_dev(e, e->d_in, count, e->d_out, e->stream);
    if (nout < 0) return nout;
    if (nout)
        HIPCHK(hipMemcpyAsync(out, e->d_out, (size_t)nout * e->ch * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIPCHK(wait_stream(e->stream));
    return nout;
}

// The device address of a pinned host buffer the kernels may store into, or nullptr (pageable memory).  Asked on
// every call (~1 us): a remembered answer could outlive the buffer it was about.
void* mapped_host_ptr(void* p) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    if (hipPointerGetAttributes(&at, p) == hipSuccess && at.type == hipMemoryTypeHost && at.devicePointer) return at.devicePointer;
    (void)hipGetLastError();   // (pageable memory: an error the runtime keeps otherwise)
    return nullptr;
}

// run() with each side on the host (pinned stream buffer) or already on the device (the
// device-resident companion of a stream): copies only where a side is on the host.
int64_t process_ex(Engine* e, const void* in, int in_dev, int count, void* out, int out_dev) {
    if (count < 0 || (count > 0 && (!in || !out))) return QDSP_HIP_EINVAL;
    // (a deferred host output is a host output: it goes through the staging buffer unless the kernel can store
    // straight into the pinned buffer)
    const bool out_host = out_dev == QDSP_HIP_LINK_HOST || out_dev == QDSP_HIP_LINK_HOST_DEFERRED;
    if ((!in_dev || out_host) && count > e->max_block) {
        int rc = ensure_io(e, count);
        if (rc) return rc;
    }
    HIPCHK(hipSetDevice(e->device));
    // in_dev / out_dev: 0 host, 1 device (complete / to be complete on return), 2 device through a pipelined link
    hipStream_t st = e->stream;
    if (in_dev == QDSP_HIP_LINK_PIPELINED || out_dev == QDSP_HIP_LINK_PIPELINED) {
        st = shared_stream(e->device);
        if (!st) return QDSP_HIP_ENOMEM;
    }
    if (e->last_stream && e->last_stream != st) HIPCHK(hipStreamSynchronize(e->last_stream));   // (links re-plumbed)
    e->last_stream = st;
    const void* src = in;
    // Host input + deferred host output into pinned memory (HandlerSource -> block -> sink on the mirror's pinned streams): see Engine::up_stream
    const bool split_upload = !in_dev && count > 0 && out_dev == QDSP_HIP_LINK_HOST_DEFERRED && e->done_ev && mapped_host_ptr(out) && mapped_host_ptr(const_cast<void*>(in)) &&
                              !qk::knob(qk::K_NO_SPLIT_UPLOAD, 0);
    if (split_upload) {
        if (!e->up_stream) {
            HIPCHK(hipStreamCreateWithFlags(&e->up_stream, hipStreamNonBlocking));
            HIPCHK(hipEventCreateWithFlags(&e->ev_up, hipEventDisableTiming));
            HIPCHK(hipEventCreateWithFlags(&e->ev_kernel, hipEventDisableTiming));
        }
        if (e->kernel_recorded) HIPCHK(hipStreamWaitEvent(e->up_stream, e->ev_kernel, 0));      // the previous kernel has read d_in
        HIPCHK(hipMemcpyAsync(e->d_in, in, (size_t)count * e->ch * sizeof(float), hipMemcpyHostToDevice, e->up_stream));
        HIPCHK(hipEventRecord(e->ev_up, e->up_stream));
        HIPCHK(hipStreamWaitEvent(st, e->ev_up, 0));
        src = e->d_in;
    } else if (!in_dev) {
        // (an earlier split upload of this handle may still be in flight on the other stream: same staging buffer)
        if (e->up_stream && e->kernel_recorded) HIPCHK(hipStreamWaitEvent(st, e->ev_up, 0));
        if (count) HIPCHK(hipMemcpyAsync(e->d_in, in, (size_t)count * e->ch * sizeof(float), hipMemcpyHostToDevice, st));
        src = e->d_in;
    }
    // A small result headed for pinned, device-mapped host memory (a decimator's output in a stream<T> buffer) is
    // stored there by the kernel itself: posted writes over the link instead of a separate copy operation, which
    // costs ~10 us of latency per call whatever its size.
    const bool deferred = out_dev == QDSP_HIP_LINK_HOST_DEFERRED;
    if (deferred) {
        if (!e->done_ev) return QDSP_HIP_EINVAL;
        out_dev = QDSP_HIP_LINK_HOST;
    }
    void* dst = out_dev ? out : e->d_out;
    bool direct_out = false;
    if (!out_dev && count > 0) {
        const size_t out_bytes = (size_t)out_size(e, count) * e->ch * sizeof(float);
        if (out_bytes > 0 && out_bytes <= (size_t)(1 << 20)) {
            void* mapped = mapped_host_ptr(out);
            if (mapped) { dst = mapped; direct_out = true; }
        }
    }
    const int64_t nout = process_dev(e, src, count, dst, st);
    if (nout < 0) return nout;
    if (split_upload) {
        HIPCHK(hipEventRecord(e->ev_kernel, st));
        e->kernel_recorded = true;
    }
    if (!out_dev && !direct_out && nout)
        HIPCHK(hipMemcpyAsync(out, e->d_out, (size_t)nout * e->ch * sizeof(float), hipMemcpyDeviceToHost, st));
    // a block handed to a pipelined link need not be complete; a host input must have left its buffer, though
    if (deferred) {
        // the consumer waits (stream<T>::read does, on the event that travels with the buffer); only a host INPUT
        // and a pageable output (whose "async" copy is not) still need this call to wait
        HIPCHK(hipEventRecord(e->done_ev, st));
        // (a device input from a producer outside the shared stream must have been read before this call returns
        // and the block flushes it: only a pipelined input lets the call go without waiting)
        if (in_dev == QDSP_HIP_LINK_PIPELINED && (direct_out || nout == 0 || mapped_host_ptr(out))) return nout;
        if (split_upload) {
            // the input buffer is the caller's again once the upload has read it; the results travel behind done_ev
            HIPCHK(hipEventSynchronize(e->ev_up));
            return nout;
        }
        HIPCHK(hipEventSynchronize(e->done_ev));
        return nout;
    }
    if (!(out_dev == QDSP_HIP_LINK_PIPELINED && in_dev == QDSP_HIP_LINK_PIPELINED))
        HIPCHK(st == e->stream ? wait_stream(st) : wait_event(e->ev0, st));
    return nout;
}

int reset(Engine* e) {
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    for (int i = 0; i < 2; i++)
        if (e->d_hist[i] && e->H > 0) HIPCHK(hipMemset(e->d_hist[i], 0, (size_t)e->H * e->ch * sizeof(float)));
    e->phase = 0;
    e->raw_valid = false;
    return 0;
}

int get_history(Engine* e, float* hist) {
    if (!hist) return QDSP_HIP_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    if (e->H > 0)
        HIPCHK(hipMemcpy(hist, e->d_hist[e->cur], (size_t)e->H * e->ch * sizeof(float), hipMemcpyDeviceToHost));
    return 0;
}
int set_history(Engine* e, const float* hist) {
    if (!hist) return QDSP_HIP_EINVAL;
    e->raw_valid = false;
    HIPCHK(hipSetDevice(e->device));
    HIPCHK(hipDeviceSynchronize());
    if (e->H > 0)
        HIPCHK(hipMemcpy(e->d_hist[e->cur], hist, (size_t)e->H * e->ch * sizeof(float), hipMemcpyHostToDevice));
    return 0;
}

// `d_hist` holds the H INPUT samples that precede the next call.  Engines with an NCO keep
// their history rotated (as the reference resampler's buffer holds the xlator's output), so
// those samples are rotated here with the phases they would have had: phase - H*dphase onward.
int set_history_dev(Engine* e, const void* d_hist, void* stream) {
    if (!d_hist) return QDSP_HIP_EINVAL;
    apply_pending_inc(e);
    e->raw_valid = false;
    HIPCHK(hipSetDevice(e->device));
    if (e->H <= 0) return 0;
    if (e->rotate && e->has_filter)
        return launch_xlate_raw(e, d_hist, e->H, e->d_hist[e->cur], e->phase - (unsigned long long)e->H * e->dphase, 0.0f,
                                static_cast<hipStream_t>(stream));
    HIPCHK(hipMemcpyAsync(e->d_hist[e->cur], d_hist, (size_t)e->H * e->ch * sizeof(float), hipMemcpyDeviceToDevice,
                          static_cast<hipStream_t>(stream)));
    return 0;
}

int get_phase(Engine* e, float* re, float* im) {
    if (!re || !im) return QDSP_HIP_EINVAL;
    double c, s;
    unit_of_fx(e->phase, 1.0L, &c, &s);
    *re = (float)c;
    *im = (float)s;
    return 0;
}
int set_phase(Engine* e, float re, float im) {
    if (re == 0.0f && im == 0.0f) return QDSP_HIP_EINVAL;
    e->raw_valid = false;
    e->phase = fx_of_turns(turns_of(re, im));
    return 0;
}

int time_process(Engine* e, const void* d_in, int64_t count, void* d_out, void* stream, int iters, float* ms) {
    if (iters <= 0 || !ms) return QDSP_HIP_EINVAL;
    HIPCHK(hipSetDevice(e->device));
    hipStream_t s = static_cast<hipStream_t>(stream);
    HIPCHK(hipEventRecord(e->ev0, s));
    for (int i = 0; i < iters; i++) {
        int64_t r = process_dev(e, d_in, count, d_out, stream);
        if (r < 0) return (int)r;
    }
    HIPCHK(hipEventRecord(e->ev1, s));
    HIPCHK(hipEventSynchronize(e->ev1));
    float t = 0.0f;
    HIPCHK(hipEventElapsedTime(&t, e->ev0, e->ev1));
    *ms = t / (float)iters;
    return 0;
}

Engine* any_engine(void* h) {
    Engine* e = static_cast<Engine*>(h);
    return (e && e->magic == kMagic) ? e : nullptr;
}

}  // namespace qh

using namespace qh;

// ==========================================================================================
// extern "C" boundary
// ==========================================================================================
extern "C" {

int qdsp_hip_abi_version(void) { return QDSP_HIP_ABI_VERSION; }

const char* qdsp_hip_error_string(int code) {
    if (code >= 0) return "success";
    switch (code) {
        case QDSP_HIP_EINVAL: return "qdsp_hip: invalid argument";
        case QDSP_HIP_ENOMEM: return "qdsp_hip: out of host memory";
        case QDSP_HIP_ESIZE: return "qdsp_hip: block larger than max_block";
        case QDSP_HIP_ENODEV: return "qdsp_hip: no usable HIP device";
        case QDSP_HIP_ERCCL: return "qdsp_hip: RCCL unavailable or an RCCL call failed (ring)";
        default: return hipGetErrorString((hipError_t)(-code));
    }
}

int qdsp_hip_device_count(int* count) {
    if (!count) return QDSP_HIP_EINVAL;
    *count = 0;
    HIPCHK(hipGetDeviceCount(count));
    return 0;
}

int qdsp_hip_device_info(int device, char* name, int name_len, char* arch, int arch_len, int* cus) {
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    if (name && name_len > 0) { strncpy(name, p.name, name_len - 1); name[name_len - 1] = 0; }
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    if (cus) *cus = p.multiProcessorCount;
    return 0;
}

int qdsp_hip_host_alloc(void** p, size_t bytes) {
    if (!p) return QDSP_HIP_EINVAL;
    HIPCHK(hipHostMalloc(p, bytes, hipHostMallocDefault));
    return 0;
}
int qdsp_hip_host_free(void* p) { HIPCHK(hipHostFree(p)); return 0; }
int qdsp_hip_host_register(void* p, size_t bytes) { HIPCHK(hipHostRegister(p, bytes, hipHostRegisterDefault)); return 0; }
int qdsp_hip_host_unregister(void* p) { HIPCHK(hipHostUnregister(p)); return 0; }
int qdsp_hip_dev_alloc(int device, void** p, size_t bytes) {
    if (!p) return QDSP_HIP_EINVAL;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMalloc(p, bytes));
    return 0;
}
int qdsp_hip_dev_free(int device, void* p) { HIPCHK(hipSetDevice(device)); HIPCHK(hipFree(p)); return 0; }
int qdsp_hip_memcpy_h2d(int device, void* d, const void* h, size_t bytes) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemcpy(d, h, bytes, hipMemcpyHostToDevice));
    return 0;
}
int qdsp_hip_memcpy_d2h(int device, void* h, const void* d, size_t bytes) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemcpy(h, d, bytes, hipMemcpyDeviceToHost));
    return 0;
}
int qdsp_hip_memcpy_d2d(int device, void* d_dst, const void* d_src, size_t bytes) {
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
    return 0;
}
// Copies for blocks that forward data between links (Splitter): ordered behind a pipelined input, not waited for
// when the destination link is pipelined too.
int qdsp_hip_memcpy_d2d_link(int device, void* d_dst, const void* d_src, size_t bytes, int in_link, int out_link) {
    HIPCHK(hipSetDevice(device));
    if (in_link != QDSP_HIP_LINK_PIPELINED && out_link != QDSP_HIP_LINK_PIPELINED) {
        HIPCHK(hipMemcpy(d_dst, d_src, bytes, hipMemcpyDeviceToDevice));
        return 0;
    }
    hipStream_t st = shared_stream(device);
    if (!st) return QDSP_HIP_ENOMEM;
    HIPCHK(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, st));
    // Only a pipelined INPUT may be released with the copy still queued (its producer launches into this same
    // stream, behind the copy).  A plain device input comes from a producer on some other stream (Add / Multiply,
    // a host-fed Splitter): it reuses the buffer two blocks later whatever this stream has got to, so the copy
    // must have read it before the caller flushes.
    if (!(in_link == QDSP_HIP_LINK_PIPELINED && out_link == QDSP_HIP_LINK_PIPELINED)) HIPCHK(hipStreamSynchronize(st));
    return 0;
}
int qdsp_hip_memcpy_d2h_link(int device, void* h_dst, const void* d_src, size_t bytes, int in_link) {
    HIPCHK(hipSetDevice(device));
    if (in_link != QDSP_HIP_LINK_PIPELINED) {
        HIPCHK(hipMemcpy(h_dst, d_src, bytes, hipMemcpyDeviceToHost));
        return 0;
    }
    hipStream_t st = shared_stream(device);
    if (!st) return QDSP_HIP_ENOMEM;
    HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    return 0;
}
int qdsp_hip_device_sync(int device) { HIPCHK(hipSetDevice(device)); HIPCHK(hipDeviceSynchronize()); return 0; }

// ---- filter-bearing operators: FIR, resampler, fused VFO ---------------------------------
#define QDSP_FILTER_COMMON(prefix, KIND)                                                          \
    int prefix##_reset(void* h) { Engine* e = as_engine(h, KIND); return e ? reset(e) : QDSP_HIP_EINVAL; } \
    int prefix##_history_len(void* h) { Engine* e = as_engine(h, KIND); return e ? e->H : QDSP_HIP_EINVAL; } \
    int prefix##_get_history(void* h, float* p) { Engine* e = as_engine(h, KIND); return e ? get_history(e, p) : QDSP_HIP_EINVAL; } \
    int prefix##_set_history(void* h, const float* p) { Engine* e = as_engine(h, KIND); return e ? set_history(e, p) : QDSP_HIP_EINVAL; } \
    int prefix##_history_dev(void* h, void** d) {                                                 \
        Engine* e = as_engine(h, KIND);                                                           \
        if (!e || !d) return QDSP_HIP_EINVAL;                                                     \
        *d = e->d_hist[e->cur];                                                                   \
        e->raw_valid = false; /* (the caller may write through the pointer) */                    \
        return 0;                                                                                 \
    }                                                                                             \
    int prefix##_set_history_dev(void* h, const void* d, void* s) {                               \
        Engine* e = as_engine(h, KIND);                                                           \
        return e ? set_history_dev(e, d, s) : QDSP_HIP_EINVAL;                                    \
    }                                                                                             \
    void prefix##_destroy(void* h) { Engine* e = as_engine(h, KIND); if (e) destroy(e); }

#define QDSP_FIR_API(prefix, CH)                                                                  \
    int prefix##_create(void** h, int device, const float* taps, int ntaps, int max_block) {      \
        int rc = create(h, KIND_FIR, device, CH, false, true, max_block);                         \
        if (rc) return rc;                                                                        \
        Engine* e = static_cast<Engine*>(*h);                                                     \
        rc = configure(e, taps, ntaps, 1, 1);                                                     \
        if (!rc) rc = ensure_io(e, max_block);                                                    \
        if (rc) { destroy(e); *h = nullptr; }                                                     \
        return rc;                                                                                \
    }                                                                                             \
    int prefix##_process(void* h, const float* in, int count, float* out) {                       \
        Engine* e = as_engine(h, KIND_FIR);                                                       \
        if (!e || e->ch != CH) return QDSP_HIP_EINVAL;                                            \
        int64_t r = process_host(e, in, count, out);                                              \
        return r < 0 ? (int)r : 0;                                                                \
    }                                                                                             \
    int prefix##_process_dev(void* h, const void* d_in, int64_t count, void* d_out, void* s) {    \
        Engine* e = as_engine(h, KIND_FIR);                                                       \
        if (!e || e->ch != CH) return QDSP_HIP_EINVAL;                                            \
        int64_t r = process_dev(e, d_in, count, d_out, s);                                        \
        return r < 0 ? (int)r : 0;                                                                \
    }                                                                                             \
    int prefix##_process_ex(void* h, const void* in, int in_dev, int count, void* out, int out_dev) { \
        Engine* e = as_engine(h, KIND_FIR);                                                       \
        if (!e || e->ch != CH) return QDSP_HIP_EINVAL;                                            \
        int64_t r = process_ex(e, in, in_dev, count, out, out_dev);                               \
        return r < 0 ? (int)r : 0;                                                                \
    }                                                                                             \
    int prefix##_set_taps(void* h, const float* taps, int ntaps) {                                \
        Engine* e = as_engine(h, KIND_FIR);                                                       \
        return e ? configure(e, taps, ntaps, 1, 1) : QDSP_HIP_EINVAL;                             \
    }                                                                                             \
    int prefix##_set_mode(void* h, int mode) {                                                    \
        Engine* e = as_engine(h, KIND_FIR);                                                       \
        if (!e || mode < 0 || mode > 2) return QDSP_HIP_EINVAL;                                   \
        e->fir_mode = mode;                                                                       \
        return 0;                                                                                 \
    }                                                                                             \
    QDSP_FILTER_COMMON(prefix, KIND_FIR)

QDSP_FIR_API(qdsp_hip_fir_cf32, 2)
QDSP_FIR_API(qdsp_hip_fir_f32, 1)

#define QDSP_DECIM_API(prefix, CH)                                                                \
    int prefix##_create(void** h, int device, const float* taps, int ntaps, int interp, int decim, int max_block) { \
        int rc = create(h, KIND_DECIM, device, CH, false, true, max_block);                       \
        if (rc) return rc;                                                                        \
        Engine* e = static_cast<Engine*>(*h);                                                     \
        rc = configure(e, taps, ntaps, interp, decim);                                            \
        if (!rc) rc = ensure_io(e, max_block);                                                    \
        if (rc) { destroy(e); *h = nullptr; }                                                     \
        return rc;                                                                                \
    }                                                                                             \
    int prefix##_process(void* h, const float* in, int count, float* out) {                       \
        Engine* e = as_engine(h, KIND_DECIM);                                                     \
        if (!e || e->ch != CH) return QDSP_HIP_EINVAL;                                            \
        return (int)process_host(e, in, count, out);                                              \
    }                                                                                             \
    int64_t prefix##_process_dev(void* h, const void* d_in, int64_t count, void* d_out, void* s) { \
        Engine* e = as_engine(h, KIND_DECIM);                                                     \
        if (!e || e->ch != CH) return QDSP_HIP_EINVAL;                                            \
        return process_dev(e, d_in, count, d_out, s);                                             \
    }                                                                                             \
    int prefix##_process_ex(void* h, const void* in, int in_dev, int count, void* out, int out_dev) { \
        Engine* e = as_engine(h, KIND_DECIM);                                                     \
        if (!e || e->ch != CH) return QDSP_HIP_EINVAL;                                            \
        return (int)process_ex(e, in, in_dev, count, out, out_dev);                               \
    }                                                                                             \
    int prefix##_configure(void* h, const float* taps, int ntaps, int interp, int decim) {        \
        Engine* e = as_engine(h, KIND_DECIM);                                                     \
        if (!e) return QDSP_HIP_EINVAL;                                                           \
        int rc = configure(e, taps, ntaps, interp, decim);                                        \
        if (!rc && e->max_block) { int mb = e->max_block; e->max_block = 0; rc = ensure_io(e, mb); } \
        return rc;                                                                                \
    }                                                                                             \
    int64_t prefix##_out_size(void* h, int64_t count) {                                           \
        Engine* e = as_engine(h, KIND_DECIM);                                                     \
        return e ? out_size(e, count) : QDSP_HIP_EINVAL;                                          \
    }                                                                                             \
    int prefix##_set_mode(void* h, int mode) {                                                    \
        Engine* e = as_engine(h, KIND_DECIM);                                                     \
        if (!e || mode < 0 || mode > 2) return QDSP_HIP_EINVAL;                                   \
        e->fir_mode = mode;                                                                       \
        return 0;                                                                                 \
    }                                                                                             \
    QDSP_FILTER_COMMON(prefix, KIND_DECIM)

QDSP_DECIM_API(qdsp_hip_decim_cf32, 2)
QDSP_DECIM_API(qdsp_hip_decim_f32, 1)

// ---- FrequencyXlator ----------------------------------------------------------------------
int qdsp_hip_xlate_cf32_create(void** h, int device, float inc_re, float inc_im, int max_block) {
    if (inc_re == 0.0f && inc_im == 0.0f) return QDSP_HIP_EINVAL;
    int rc = create(h, KIND_XLATE, device, 2, true, false, max_block);
    if (rc) return rc;
    Engine* e = static_cast<Engine*>(*h);
    set_inc_now(e, inc_re, inc_im);
    rc = ensure_io(e, max_block);
    if (rc) { destroy(e); *h = nullptr; }
    return rc;
}
int qdsp_hip_xlate_cf32_process(void* h, const float* in, int count, float* out) {
    Engine* e = as_engine(h, KIND_XLATE);
    if (!e) return QDSP_HIP_EINVAL;
    int64_t r = process_host(e, in, count, out);
    return r < 0 ? (int)r : 0;
}
int qdsp_hip_xlate_cf32_process_dev(void* h, const void* d_in, int64_t count, void* d_out, void* s) {
    Engine* e = as_engine(h, KIND_XLATE);
    if (!e) return QDSP_HIP_EINVAL;
    int64_t r = process_dev(e, d_in, count, d_out, s);
    return r < 0 ? (int)r : 0;
}
int qdsp_hip_xlate_cf32_process_ex(void* h, const void* in, int in_dev, int count, void* out, int out_dev) {
    Engine* e = as_engine(h, KIND_XLATE);
    if (!e) return QDSP_HIP_EINVAL;
    int64_t r = process_ex(e, in, in_dev, count, out, out_dev);
    return r < 0 ? (int)r : 0;
}
int qdsp_hip_xlate_cf32_set_phase_inc(void* h, float re, float im) {
    Engine* e = as_engine(h, KIND_XLATE);
    if (!e || (re == 0.0f && im == 0.0f)) return QDSP_HIP_EINVAL;
    set_inc(e, re, im);
    return 0;
}
int qdsp_hip_xlate_cf32_get_phase(void* h, float* re, float* im) {
    Engine* e = as_engine(h, KIND_XLATE);
    return e ? get_phase(e, re, im) : QDSP_HIP_EINVAL;
}
int qdsp_hip_xlate_cf32_set_phase(void* h, float re, float im) {
    Engine* e = as_engine(h, KIND_XLATE);
    return e ? set_phase(e, re, im) : QDSP_HIP_EINVAL;
}
int qdsp_hip_xlate_cf32_advance(void* h, int64_t n) {
    Engine* e = as_engine(h, KIND_XLATE);
    if (!e) return QDSP_HIP_EINVAL;
    apply_pending_inc(e);
    e->phase += (unsigned long long)n * e->dphase;
    e->raw_valid = false;
    return 0;
}
int qdsp_hip_xlate_cf32_set_volk_gain(void* h, int on) {
    Engine* e = as_engine(h, KIND_XLATE);
    if (!e) return QDSP_HIP_EINVAL;
    e->volk_gain = on != 0;
    return 0;
}
void qdsp_hip_xlate_cf32_destroy(void* h) { Engine* e = as_engine(h, KIND_XLATE); if (e) destroy(e); }

// ---- SineSource -----------------------------------------------------------------------------
int qdsp_hip_sine_cf32_create(void** h, int device, float inc_re, float inc_im, int max_block) {
    if (inc_re == 0.0f && inc_im == 0.0f) return QDSP_HIP_EINVAL;
    int rc = create(h, KIND_SINE, device, 2, true, false, max_block);
    if (rc) return rc;
    Engine* e = static_cast<Engine*>(*h);
    set_inc_now(e, inc_re, inc_im);
    rc = ensure_io(e, max_block);
    if (rc) { destroy(e); *h = nullptr; }
    return rc;
}
int qdsp_hip_sine_cf32_generate(void* h, int count, void* out, int out_on_device) {
    Engine* e = as_engine(h, KIND_SINE);
    if (!e || count < 0 || (count > 0 && !out)) return QDSP_HIP_EINVAL;
    if ((out_on_device == QDSP_HIP_LINK_HOST || out_on_device == QDSP_HIP_LINK_HOST_DEFERRED) && count > e->max_block) {
        int rc = ensure_io(e, count);
        if (rc) return rc;
    }
    HIPCHK(hipSetDevice(e->device));
    hipStream_t st = e->stream;
    if (out_on_device == QDSP_HIP_LINK_PIPELINED) {
        st = shared_stream(e->device);
        if (!st) return QDSP_HIP_ENOMEM;
    }
    if (e->last_stream && e->last_stream != st) HIPCHK(hipStreamSynchronize(e->last_stream));
    e->last_stream = st;
    const bool deferred = out_on_device == QDSP_HIP_LINK_HOST_DEFERRED;
    if (deferred) {
        if (!e->done_ev) return QDSP_HIP_EINVAL;
        out_on_device = QDSP_HIP_LINK_HOST;
    }
    void* dst = out_on_device ? out : e->d_out;
    const int64_t r = process_dev(e, nullptr, count, dst, st);
    if (r < 0) return (int)r;
    if (!out_on_device && count)
        HIPCHK(hipMemcpyAsync(out, e->d_out, (size_t)count * 8, hipMemcpyDeviceToHost, st));
    if (deferred) {
        HIPCHK(hipEventRecord(e->done_ev, st));
        if (count == 0 || mapped_host_ptr(out)) return 0;
        HIPCHK(hipEventSynchronize(e->done_ev));
        return 0;
    }
    if (out_on_device != QDSP_HIP_LINK_PIPELINED) HIPCHK(wait_stream(st));
    return 0;
}
int qdsp_hip_sine_cf32_generate_dev(void* h, int64_t count, void* d_out, void* stream) {
    Engine* e = as_engine(h, KIND_SINE);
    if (!e) return QDSP_HIP_EINVAL;
    const int64_t r = process_dev(e, nullptr, count, d_out, stream);
    return r < 0 ? (int)r : 0;
}
int qdsp_hip_sine_cf32_set_phase_inc(void* h, float re, float im) {
    Engine* e = as_engine(h, KIND_SINE);
    if (!e || (re == 0.0f && im == 0.0f)) return QDSP_HIP_EINVAL;
    set_inc(e, re, im);
    return 0;
}
int qdsp_hip_sine_cf32_get_phase(void* h, float* re, float* im) {
    Engine* e = as_engine(h, KIND_SINE);
    return e ? get_phase(e, re, im) : QDSP_HIP_EINVAL;
}
int qdsp_hip_sine_cf32_set_volk_gain(void* h, int on) {
    Engine* e = as_engine(h, KIND_SINE);
    if (!e) return QDSP_HIP_EINVAL;
    e->volk_gain = on != 0;
    return 0;
}
void qdsp_hip_sine_cf32_destroy(void* h) { Engine* e = as_engine(h, KIND_SINE); if (e) destroy(e); }

// ---- fused VFO ------------------------------------------------------------------------------
int qdsp_hip_xlate_fir_decim_cf32_create(void** h, int device, const float* taps, int ntaps, int interp,
                                         int decim, float inc_re, float inc_im, int max_block) {
    if (inc_re == 0.0f && inc_im == 0.0f) return QDSP_HIP_EINVAL;
    int rc = create(h, KIND_VFO, device, 2, true, true, max_block);
    if (rc) return rc;
    Engine* e = static_cast<Engine*>(*h);
    set_inc_now(e, inc_re, inc_im);
    rc = configure(e, taps, ntaps, interp, decim);
    if (!rc) rc = ensure_io(e, max_block);
    if (rc) { destroy(e); *h = nullptr; }
    return rc;
}
int qdsp_hip_xlate_fir_decim_cf32_process(void* h, const float* in, int count, float* out) {
    Engine* e = as_engine(h, KIND_VFO);
    return e ? (int)process_host(e, in, count, out) : QDSP_HIP_EINVAL;
}
int64_t qdsp_hip_xlate_fir_decim_cf32_process_dev(void* h, const void* d_in, int64_t count, void* d_out, void* s) {
    Engine* e = as_engine(h, KIND_VFO);
    return e ? process_dev(e, d_in, count, d_out, s) : QDSP_HIP_EINVAL;
}
int qdsp_hip_xlate_fir_decim_cf32_process_ex(void* h, const void* in, int in_dev, int count, void* out, int out_dev) {
    Engine* e = as_engine(h, KIND_VFO);
    return e ? (int)process_ex(e, in, in_dev, count, out, out_dev) : QDSP_HIP_EINVAL;
}
int qdsp_hip_xlate_fir_decim_cf32_configure(void* h, const float* taps, int ntaps, int interp, int decim) {
    Engine* e = as_engine(h, KIND_VFO);
    if (!e) return QDSP_HIP_EINVAL;
    int rc = configure(e, taps, ntaps, interp, decim);
    if (!rc && e->max_block) { int mb = e->max_block; e->max_block = 0; rc = ensure_io(e, mb); }
    return rc;
}
int qdsp_hip_xlate_fir_decim_cf32_set_phase_inc(void* h, float re, float im) {
    Engine* e = as_engine(h, KIND_VFO);
    if (!e || (re == 0.0f && im == 0.0f)) return QDSP_HIP_EINVAL;
    set_inc(e, re, im);
    return 0;
}
int qdsp_hip_xlate_fir_decim_cf32_get_phase(void* h, float* re, float* im) {
    Engine* e = as_engine(h, KIND_VFO);
    return e ? get_phase(e, re, im) : QDSP_HIP_EINVAL;
}
int qdsp_hip_xlate_fir_decim_cf32_set_phase(void* h, float re, float im) {
    Engine* e = as_engine(h, KIND_VFO);
    return e ? set_phase(e, re, im) : QDSP_HIP_EINVAL;
}
int qdsp_hip_xlate_fir_decim_cf32_advance(void* h, int64_t n) {
    Engine* e = as_engine(h, KIND_VFO);
    if (!e) return QDSP_HIP_EINVAL;
    apply_pending_inc(e);
    e->phase += (unsigned long long)n * e->dphase;
    e->raw_valid = false;
    return 0;
}
int qdsp_hip_xlate_fir_decim_cf32_set_volk_gain(void* h, int on) {
    Engine* e = as_engine(h, KIND_VFO);
    if (!e) return QDSP_HIP_EINVAL;
    e->volk_gain = on != 0;
    return 0;
}
int qdsp_hip_xlate_fir_decim_cf32_set_mode(void* h, int mode) {
    Engine* e = as_engine(h, KIND_VFO);
    if (!e || mode < 0 || mode > 2) return QDSP_HIP_EINVAL;
    e->fir_mode = mode;
    return 0;
}
int64_t qdsp_hip_xlate_fir_decim_cf32_out_size(void* h, int64_t count) {
    Engine* e = as_engine(h, KIND_VFO);
    return e ? out_size(e, count) : QDSP_HIP_EINVAL;
}
QDSP_FILTER_COMMON(qdsp_hip_xlate_fir_decim_cf32, KIND_VFO)

}  // extern "C"
