// Unified MIDI token codec on the device (the step either side of the model: SURVEY §8(f) N3).
//   decode: sampled token rows -> MidiStream events (note | -1, velocity, duration ticks) + absolute times
//           = UnifiedReconstruct.reconstruct_stream (data/transforms.py:339-365) with unflatten_notes_
//           (transforms.py:94-128) and the time scan of stream_to_raw_stream_ (data/parse_midi_numba.py:196-204)
//   encode: (note, velocity, duration-bin) events -> padded token rows
//           = flatten_notes (transforms.py:11-38) + window (172-176) + UnifiedEncoding.pad (149-152)
// Integer work, HBM-trivial: one workgroup (1024 threads) per row walks the row in 1024-element chunks with
// block-wide exclusive scans (wave shuffles + one LDS pass) and running carries; no workspace, no atomics.
#include "mgx_common.h"

namespace {

constexpr int CT = 1024;     // threads per row

// exclusive scan of one int per thread over the block; returns the thread's offset, *total = block sum
__device__ __forceinline__ int block_scan(int v, int* total, int* lds /*[17]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, 64);
        if (lane >= o) inc += t;
    }
    __syncthreads();                       // lds may still be read from the previous scan
    if (lane == 63) lds[w] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int s = 0;
        for (int i = 0; i < CT / 64; ++i) {
            const int t = lds[i];
            lds[i] = s;
            s += t;
        }
        lds[16] = s;
    }
    __syncthreads();
    *total = lds[16];
    return lds[w] + inc - v;
}

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return min(max(x, lo), hi); }

// tokens [B,L] int32 -> events [B,L,3] int32 (first counts[b] rows valid), abs_time [B,L] int32, counts [B].
// counts[b] = -1: the row holds no note-class token (the reference raises IndexError there).
__global__ __launch_bounds__(CT) void unified_decode_kernel(const int* __restrict__ tokens, int L,
                                                             const int* __restrict__ dur_ticks, int note_bins,
                                                             int velocity_bins, int duration_bins,
                                                             int* __restrict__ events, int* __restrict__ abs_time,
                                                             int* __restrict__ counts) {
    __shared__ int lds[17];
    __shared__ int s_first0, s_first1, s_lo;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int* tk = tokens + (long)b * L;
    int* ev = events + (long)b * L * 3;
    int* at = abs_time + (long)b * L;
    const int shift_v = note_bins, shift_nd = note_bins + velocity_bins, shift_d = shift_nd + duration_bins;

    // -- padding (transforms.py:346-350): any 0 -> drop element 0; any 1 (after that) -> cut at the first 1
    if (tid == 0) {
        s_first0 = L;
        s_first1 = L;
        s_lo = L;
    }
    __syncthreads();
    int m0 = L;
    for (int i = tid; i < L; i += CT)
        if (tk[i] == 0) m0 = min(m0, i);
    if (m0 < L) atomicMin(&s_first0, m0);
    __syncthreads();
    const int beg = s_first0 < L ? 1 : 0;
    int m1 = L;
    for (int i = beg + tid; i < L; i += CT)
        if (tk[i] == 1) m1 = min(m1, i);
    if (m1 < L) atomicMin(&s_first1, m1);
    __syncthreads();
    const int end = s_first1;                                     // tokens [beg, end) survive, then -2
    // -- head trim (transforms.py:105-108): first element below note_bins
    int ml = L;
    for (int i = beg + tid; i < end; i += CT)
        if (tk[i] - 2 < note_bins) ml = min(ml, i);
    if (ml < L) atomicMin(&s_lo, ml);
    __syncthreads();
    const int lo = s_lo;
    if (lo >= end) {
        if (tid == 0) counts[b] = -1;
        return;
    }
    // -- pass 1: the three streams n / v / d grow independently (a time shift appends to all three); element k
    //    of stream s is parked at ev[k][s] (positions never exceed the token index: in-place is safe because
    //    each chunk reads its tokens before any thread writes, and tokens live in another buffer anyway)
    int cn = 0, cv = 0, cd = 0;
    for (int c0 = lo; c0 < end; c0 += CT) {
        const int i = c0 + tid;
        int x = 0, an = 0, av = 0, ad = 0, vn = 0, vv = 0, vd = 0;
        if (i < end) {
            x = tk[i] - 2;
            if (x < note_bins) {
                an = 1;
                vn = clampi(x, 0, note_bins - 1);
            } else if (x < shift_nd) {
                av = 1;
                vv = clampi(x - shift_v, 0, velocity_bins - 1);
            } else if (x < shift_d) {
                ad = 1;
                vd = clampi(x - shift_nd, 0, duration_bins - 1);
            } else {
                an = av = ad = 1;
                vd = clampi(x - shift_d, 0, duration_bins - 1);
            }
        }
        int tn, tv, td;
        const int on = block_scan(an, &tn, lds);
        const int ov = block_scan(av, &tv, lds);
        const int od = block_scan(ad, &td, lds);
        if (an) ev[3L * (cn + on) + 0] = vn;
        if (av) ev[3L * (cv + ov) + 1] = vv;
        if (ad) ev[3L * (cd + od) + 2] = vd;
        cn += tn;
        cv += tv;
        cd += td;
    }
    __syncthreads();
    const int trunc = min(cn, min(cv, cd));                       // transforms.py:123-127
    // -- pass 2: note -= 1, drop velocity-0 notes, unbin the duration, running absolute time; compaction in
    //    place (a chunk's triples are in registers before the barrier inside block_scan; it writes at or
    //    below its own positions, never into a later chunk's unread range)
    int kept = 0, tbase = 0;
    for (int c0 = 0; c0 < trunc; c0 += CT) {
        const int k = c0 + tid;
        int n = 0, v = 0, d = 0, keep = 0, dt = 0;
        if (k < trunc) {
            n = ev[3L * k + 0] - 1;
            v = ev[3L * k + 1];
            d = dur_ticks[ev[3L * k + 2]];
            keep = (v != 0) | (n == -1);
            dt = (keep && n == -1) ? d : 0;                       // parse_midi_numba.py:197-199
        }
        int tk_, tt;
        const int ok = block_scan(keep, &tk_, lds);
        const int ot = block_scan(dt, &tt, lds);
        if (keep) {
            const long o = kept + ok;
            ev[3 * o + 0] = n;
            ev[3 * o + 1] = v;
            ev[3 * o + 2] = d;
            at[o] = tbase + ot;                                   // time BEFORE this event (a shift adds after)
        }
        kept += tk_;
        tbase += tt;
    }
    if (tid == 0) counts[b] = kept;
}

// events [B,Nmax] (notes 0 = time shift, velocities, binned durations), lengths [B], starts [B] -> tokens [B,seq_len]
__global__ __launch_bounds__(CT) void unified_encode_kernel(const int* __restrict__ notes, const int* __restrict__ vels,
                                                             const int* __restrict__ durs, const int* __restrict__ lengths,
                                                             const int* __restrict__ starts, int Nmax, int seq_len,
                                                             int note_bins, int velocity_bins, int duration_bins,
                                                             int* __restrict__ tokens) {
    __shared__ int lds[17];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n_ev = min(lengths[b], Nmax);
    const int* nn = notes + (long)b * Nmax;
    const int* vv = vels + (long)b * Nmax;
    const int* dd = durs + (long)b * Nmax;
    int* out = tokens + (long)b * seq_len;
    const int shift_v = note_bins, shift_nd = note_bins + velocity_bins, shift_d = shift_nd + duration_bins;
    // flattened length first (the window only applies when it exceeds seq_len - 2: transforms.py:173)
    int part = 0;
    for (int i = tid; i < n_ev; i += CT) part += nn[i] != 0 ? 3 : 1;
    int flat_len;
    block_scan(part, &flat_len, lds);
    const int win = seq_len - 2;
    const int w0 = flat_len > win ? starts[b] : 0;
    const int w1 = flat_len > win ? w0 + win : flat_len;          // flattened positions [w0, w1) are kept
    for (int i = tid; i < seq_len; i += CT) out[i] = i == 0 ? 0 : 1;     // BOS, then EOS / padding (transforms.py:152)
    __syncthreads();
    int carry = 0;
    for (int c0 = 0; c0 < n_ev; c0 += CT) {
        const int i = c0 + tid;
        int len = 0, n = 0;
        if (i < n_ev) {
            n = nn[i];
            len = n != 0 ? 3 : 1;
        }
        int tot;
        const int pos = carry + block_scan(len, &tot, lds);
        if (i < n_ev) {
            int t[3];
            if (n == 0) {
                t[0] = shift_d + dd[i];
            } else {
                t[0] = n;
                t[1] = shift_v + vv[i];
                t[2] = shift_nd + dd[i];
            }
            for (int k = 0; k < len; ++k) {
                const int p = pos + k;
                if (p >= w0 && p < w1) out[1 + p - w0] = t[k] + 2;
            }
        }
        carry += tot;
    }
}

// Separated stream: tokens [B,3,L] (note / velocity / duration rows) -> events, abs_time, counts as above
// (SeparatedReconstruct.reconstruct_stream, data/transforms.py:290-318).
__global__ __launch_bounds__(CT) void separated_decode_kernel(const int* __restrict__ tokens, int L,
                                                               const int* __restrict__ dur_ticks, int note_bins,
                                                               int velocity_bins, int duration_bins,
                                                               int* __restrict__ events, int* __restrict__ abs_time,
                                                               int* __restrict__ counts) {
    __shared__ int lds[17];
    __shared__ int s_any0, s_any1, s_first1n;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int* tn = tokens + (long)b * 3 * L;
    const int* tv = tn + L;
    const int* td = tn + 2 * L;
    int* ev = events + (long)b * L * 3;
    int* at = abs_time + (long)b * L;
    if (tid == 0) {
        s_any0 = 0;
        s_any1 = 0;
        s_first1n = L;
    }
    __syncthreads();
    int a0 = 0;
    for (int i = tid; i < L; i += CT) a0 |= (tn[i] == 0) | (tv[i] == 0) | (td[i] == 0);
    if (a0) s_any0 = 1;                                           // benign race: every writer stores 1
    __syncthreads();
    const int beg = s_any0 ? 1 : 0;                               // transforms.py:297-298
    int a1 = 0, m1 = L;
    for (int i = beg + tid; i < L; i += CT) {
        a1 |= (tn[i] == 1) | (tv[i] == 1) | (td[i] == 1);
        if (tn[i] == 1) m1 = min(m1, i);
    }
    if (a1) s_any1 = 1;
    if (m1 < L) atomicMin(&s_first1n, m1);
    __syncthreads();
    // any 1 in ANY row cuts at the first 1 of the NOTE row; argmax of an all-false row is 0: nothing survives
    const int end = s_any1 ? (s_first1n < L ? s_first1n : beg) : L;
    int kept = 0, tbase = 0;
    for (int c0 = beg; c0 < end; c0 += CT) {
        const int i = c0 + tid;
        int n = 0, v = 0, d = 0, keep = 0, dt = 0;
        if (i < end) {
            n = clampi(tn[i] - 2, 0, note_bins) - 1;
            v = clampi(tv[i] - 2, 0, velocity_bins);
            d = dur_ticks[clampi(td[i] - 2, 0, duration_bins - 1)];
            keep = (v != 0) | (n == -1);
            dt = (keep && n == -1) ? d : 0;
        }
        int tk_, tt;
        const int ok = block_scan(keep, &tk_, lds);
        const int ot = block_scan(dt, &tt, lds);
        if (keep) {
            const long o = kept + ok;
            ev[3 * o + 0] = n;
            ev[3 * o + 1] = v;
            ev[3 * o + 2] = d;
            at[o] = tbase + ot;
        }
        kept += tk_;
        tbase += tt;
    }
    if (tid == 0) counts[b] = kept;
}

}  // namespace

extern "C" int mgx_separated_decode_events(const int* tokens, int B, int L, const int* dur_ticks, int note_bins,
                                           int velocity_bins, int duration_bins, int* events, int* abs_time,
                                           int* counts, void* stream) {
    MGX_REQUIRE(tokens && dur_ticks && events && abs_time && counts, MGX_EINVAL, "mgx_separated_decode_events: null pointer");
    MGX_REQUIRE(B > 0 && L > 0, MGX_EINVAL, "mgx_separated_decode_events: B=%d L=%d", B, L);
    MGX_REQUIRE(note_bins > 0 && velocity_bins > 0 && duration_bins > 0, MGX_EINVAL,
                "mgx_separated_decode_events: bins %d/%d/%d", note_bins, velocity_bins, duration_bins);
    hipLaunchKernelGGL(separated_decode_kernel, dim3(B), dim3(CT), 0, (hipStream_t)stream, tokens, L, dur_ticks,
                       note_bins, velocity_bins, duration_bins, events, abs_time, counts);
    MGX_CHECK_LAUNCH("separated_decode");
    return MGX_OK;
}

extern "C" int mgx_unified_decode_events(const int* tokens, int B, int L, const int* dur_ticks, int note_bins,
                                         int velocity_bins, int duration_bins, int* events, int* abs_time, int* counts,
                                         void* stream) {
    MGX_REQUIRE(tokens && dur_ticks && events && abs_time && counts, MGX_EINVAL, "mgx_unified_decode_events: null pointer");
    MGX_REQUIRE(B > 0 && L > 0, MGX_EINVAL, "mgx_unified_decode_events: B=%d L=%d", B, L);
    MGX_REQUIRE(note_bins > 0 && velocity_bins > 0 && duration_bins > 0, MGX_EINVAL,
                "mgx_unified_decode_events: bins %d/%d/%d", note_bins, velocity_bins, duration_bins);
    hipLaunchKernelGGL(unified_decode_kernel, dim3(B), dim3(CT), 0, (hipStream_t)stream, tokens, L, dur_ticks, note_bins,
                       velocity_bins, duration_bins, events, abs_time, counts);
    MGX_CHECK_LAUNCH("unified_decode");
    return MGX_OK;
}

extern "C" int mgx_unified_encode_tokens(const int* notes, const int* velocities, const int* durations,
                                         const int* lengths, const int* starts, int B, int Nmax, int seq_len,
                                         int note_bins, int velocity_bins, int duration_bins, int* tokens,
                                         void* stream) {
    MGX_REQUIRE(notes && velocities && durations && lengths && starts && tokens, MGX_EINVAL,
                "mgx_unified_encode_tokens: null pointer");
    MGX_REQUIRE(B > 0 && Nmax > 0 && seq_len >= 2, MGX_EINVAL, "mgx_unified_encode_tokens: B=%d Nmax=%d seq_len=%d", B,
                Nmax, seq_len);
    MGX_REQUIRE((long)Nmax * 3 < (1L << 30), MGX_EUNSUPPORTED, "mgx_unified_encode_tokens: Nmax=%d too large", Nmax);
    hipLaunchKernelGGL(unified_encode_kernel, dim3(B), dim3(CT), 0, (hipStream_t)stream, notes, velocities, durations,
                       lengths, starts, Nmax, seq_len, note_bins, velocity_bins, duration_bins, tokens);
    MGX_CHECK_LAUNCH("unified_encode");
    return MGX_OK;
}
