// Conv weight gradients of the whole CNN step in ONE launch - round 6: TAP-SHARED operand tiles.
//
//     dW[tap][c][n] = sum_m X[m + tap - 1][c] * dZ[m][n]      (0 when level(m) + tap - 1 leaves the column)
//
// conv_wgrad2.h folds the taps into one contraction-free axis kk = tap * kpt + c and streams, per 32-row slab, a [32][256] tile of
// the SHIFTED rows H'[m][kk] = X[m + tap - 1][c] beside the [32][256] dZ tile: 32 KB per slab, and the three taps of a channel are
// three different tiles (other workgroups, or other rows of this one).  Its stamps (profiles/r05_cnn_wgrad_stamps.txt): the operand
// requests alone take the whole loop (1561 of 1596 clocks per slab against 896 of MFMAs) at 21 B/clk per CU, limited at the same
// level by the loader waves' ~25 instructions per 1-KiB piece (per-lane source selection for out-of-column levels) and by the
// memory side.  Here:
//   * a tile of a 3-tap conv is 16 GROUP SLOTS of 16 kk rows = 3 taps x 5 channel blocks of 16 (80 channels) + one spare slot; its
//     X operand is the UNSHIFTED rows m0 - 1 .. m0 + 32 (34 rows) of those channels, fetched ONCE for the three taps - the tap is a
//     row offset of the transposing LDS read.  406 channels = 5 tiles x 80 + 6: the sixth 16-block rides in the spare slots of tiles
//     1..3 (one tap each, 16 more channels in their X tile), the spare slot of tile 0 is the column of ones whose product row is the
//     bias gradient.  Per slab: X 34 x 96 channels (6.4 KB) + dZ 32 x 224 (14 KB) = 20.4 KB instead of 32;
//   * rows whose shifted level leaves the column (level 0 under tap 0, level seq - 1 under tap 2) cannot be zeroed in the tile -
//     the same row is valid under the other taps - so the READING lane is pointed at 8 zero bytes instead (a per-lane key per group
//     slot, one compare + one select per read);
//   * the loaders' fast path is pointer += constant: every lane's source is decided once per queue entry (real channels, zero page,
//     ones page); the per-lane row tests run only in the first slab of the batch and from the last one on;
//   * a slab stays in the ring while it is computed on and the seven dZ column groups stream through four fragment buffers (12 MFMAs
//     between request and use) instead of all being fetched a slab ahead: 16 VGPRs less - no scratch access inside the loop at 168
//     VGPRs (conv_wgrad2.h: 184 B per lane) - and the 20-KB slabs make room for a FIVE-slot ring;
//   * 1-tap (residual) convs use the same code with 16 windows of 16 channels (the old tile), shift 0, four slots.
// Everything else - 8 compute waves of 64(kk) x 112(n) (4 x 7 v_mfma_f32_16x16x32_bf16), 4 loader waves, LDS-DMA ring, one persistent
// workgroup per CU taking (tile, row range) entries from eight host-built queues, partial sums out by float atomics through a staging
// area - is conv_wgrad2.h's (hpo_train.py:159-200 is what the gradients belong to).
//
// Measured (profiles/r06_cnn_wgrad3.txt, batch 512, stamps in shader clocks per 32-row slab of a 3-tap tile; 896 = the MFMAs alone):
// k_conv_wgrad2l 1596 -> 1357 here; kernel 0.915 -> ~0.80 ms, CNN step 3.17 -> 3.07 ms.  What is left, by ablation builds: MFMAs +
// barrier 954; + the 22 fragment reads and their 17 address operations 1179; + the boundary selects (16 VALU) 1335 - a VALU
// operation costs ~8 clocks here however it is placed or branched around (two waves per SIMD run the same code in step) - and the
// operand requests ALONE take 1320 (566 .. 1876 from entry to entry): 20.4 KB per slab arrive at 15.5 B/clk per CU = 7.9 TB/s over the
// chip through the L2s at 57 % hits.  More requests in flight make it slower, not faster (an L2 prefetch 4 / 8 slabs ahead: 1594 /
// 2182 clocks per slab): the memory side is at its throughput for this access pattern, so a faster compute loop alone buys nothing.
#pragma once
#include "conv_wgrad2.h"
#ifndef CW3_EXP
#define CW3_EXP 0                // development (timing only): 1 = no sched_barriers in the loop, 2 = no column-boundary redirect (wrong sums), 8 = no per-slab
#endif                           // barriers (races), 16 = the redirect's instructions with keys that never match (wrong sums)
#define CW3_BAR() do { if (!(CW3_EXP & 8)) __builtin_amdgcn_s_barrier(); } while (0)
#define CW3_SB() do { if (!(CW3_EXP & 1)) __builtin_amdgcn_sched_barrier(0); } while (0)

struct Cw3Tile {
    const u16* H; const u16* Z;
    float* dW; float* db;
    int ldh, ldz, cin, cout, n0;
    int ntap;                    // 1 | 3
    int zchunks;                 // 16-B chunks of a dZ row that carry columns of this tile (the rest comes from the zero page)
    short win_c[16];             // X-tile window j (16 channels) -> first channel, -1 = zeros, -2 = ones (first element of the window)
    signed char slot_tap[16];    // group slot -> tap, -1 = empty
    signed char slot_win[16];    // group slot -> window
};
struct Cw3Args {
    const Cw3Tile* tiles; int n_tiles;
    const CwWork* work; int q_begin[9]; int* counters;       // as CwArgs
    int64_t m_rows; int seq;
    const u16* zeros;
    unsigned long long* dbg;
};

// LDS: a ring of slabs [X tile | dZ tile].  3-tap tiles: 8 KB (6.4 used) + 16 KB, FIVE slots; 1-tap tiles: 16 + 16 KB, four slots.  A
// slab stays in the ring while it is computed on (the dZ fragments are streamed from it three column groups ahead of the MFMAs
// instead of all 28 registers' worth being fetched a slab ahead: 16 VGPRs less, which is what ends the spills of conv_wgrad2.h at
// 168 VGPRs), so the loaders run SLOTS - 2 slabs ahead of the one in use: 3 (as conv_wgrad2.h) and 2.
// Float atomic on a pointer KNOWN to be global memory.  `atomicAdd(float*)` on a pointer that came out of a descriptor in memory is a
// FLAT atomic: it counts in lgkmcnt as well as vmcnt, so every LDS read of the flush behind it waits for the atomic's address to be
// resolved (round 6, read off the compiled flush: 160 flat_atomic_add_f32 per wave and entry, each followed by staging reads).
__device__ __forceinline__ void cw3_atomic_add(float* p, float v) {
    typedef float __attribute__((address_space(1))) * gptr;
    (void)__builtin_amdgcn_global_atomic_fadd_f32((gptr)p, v);
}

template <int W> struct Cw3Geo {
    static constexpr int SLOTS = W == 6 ? 5 : 4;
    static constexpr unsigned XB = W == 6 ? 8192u : 16384u;          // bytes of the X tile's region = offset of the dZ tile
    static constexpr unsigned SLAB = XB + 16384u;
};
#define CW3_STAGE_OFF 131072                                         // flush staging behind the larger of the two rings (5 x 24 KB, 4 x 32 KB)
#define CW3_QWORD_OFF (CW3_STAGE_OFF + CW2L_STAGE_BYTES)             // [2] queue words
#define CW3_ZERO_OFF (CW3_QWORD_OFF + 16)                            // 16 zero bytes
#define CW3_LDS_BYTES (CW3_ZERO_OFF + 16)

// Byte offset of element (row r, window j) in an X tile of W windows per row.  Rows r, r+1, r+2, r+3 and r+8 .. r+11 - what one
// half-wave of ds_read_b64_tr_b16 touches - land on eight different 32-byte bank groups for EVERY r (the tap shifts the rows):
//   W = 6  (192-B rows): bank group = (6 r + (j ^ bit3(r))) mod 8: 6 r gives the four even groups, bit 3 of r the parity;
//   W = 16 (512-B rows): bank group = (j ^ (2 (r & 3) + bit3(r))) mod 8.
template <int W>
__device__ __forceinline__ int cw3_sw(int r) { return W == 6 ? ((r >> 3) & 1) : (((r & 3) << 1) | ((r >> 3) & 1)); }
template <int W>
__device__ __forceinline__ int cw3_xoff(int r, int j) { return r * (W * 32) + ((j ^ cw3_sw<W>(r)) << 5); }

__device__ __forceinline__ void cw3_stamp_head(const Cw3Args& pa, int tid) {
    if (pa.dbg && tid == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        pa.dbg[(int64_t)blockIdx.x * CW_DBG_SLOTS] = __builtin_amdgcn_s_memrealtime();
        pa.dbg[(int64_t)blockIdx.x * CW_DBG_SLOTS + 1] = ((unsigned long long)(xcc & 0xf) << 32) | hw;
    }
}
__device__ __forceinline__ void cw3_stamp(const Cw3Args& pa, int tid, int& slot, int tag) {
    if (pa.dbg) {
        if (tid == 0 && slot < CW_DBG_SLOTS - 1)
            pa.dbg[(int64_t)blockIdx.x * CW_DBG_SLOTS + slot] = ((unsigned long long)tag << 48) | (__builtin_amdgcn_s_memtime() & 0xffffffffffffull);
        ++slot;
    }
}

// ---- loader wave: NX pieces of the X tile + 4 of the dZ tile per slab.
// X piece q (= NX lw + jj) is bytes [1024 q, 1024 q + 1024) of the tile: lane -> 16-B chunk id 64 q + lane = (row, physical half-window).
template <int W, int R, int NX>
__device__ __forceinline__ void cw3_loader(const Cw3Args& pa, const Cw3Tile* __restrict__ Tg, const Cw3Tile& T, int s0, int nsl, int lw, int lane,
                                           unsigned lds0, int q, int par, int* next_slot, bool fetch) {
    constexpr int NI = NX + 4;
    constexpr int HALO = W == 6 ? 1 : 0;
    constexpr int SLOTS = Cw3Geo<W>::SLOTS;
    constexpr unsigned XB = Cw3Geo<W>::XB, SLAB = Cw3Geo<W>::SLAB;
    const char* zpage = reinterpret_cast<const char*>(pa.zeros);
    const char* opage = zpage + 64;                                  // {1.0, 0, 0, 0, 0, 0, 0, 0} bf16
    const int64_t ldh2 = (int64_t)T.ldh * 2, ldz2 = (int64_t)T.ldz * 2;
    const int m_rows = (int)pa.m_rows;
    const char* xp[NX]; const char* zp[4];      // source of this lane's chunk in the next slab to issue
    unsigned xinc[NX], zinc[4];                 // bytes per slab: 32 rows, or 0 for a lane that reads a constant page
    int xr[NX];                                 // tile row of the chunk, -1 = the lane never reads the tensor
#pragma unroll
    for (int jj = 0; jj < NX; ++jj) {
        const int id = (NX * lw + jj) * 64 + lane;
        const int r = id / (2 * W), p = id - r * (2 * W);
        const int j = (p >> 1) ^ cw3_sw<W>(r);
        int c = -1;
        if (r < R) c = Tg->win_c[j];
        if (c >= 0) {
            xp[jj] = reinterpret_cast<const char*>(T.H + c + (p & 1) * 8) + (int64_t)(s0 * 32 + r - HALO) * ldh2;
            xinc[jj] = (unsigned)(32 * ldh2); xr[jj] = r;
        } else {
            xp[jj] = (c == -2 && (p & 1) == 0) ? opage : zpage;
            xinc[jj] = 0u; xr[jj] = -1;
        }
    }
    int zr[4];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        const int r = 2 * (4 * lw + jj) + (lane >> 5), pch = lane & 31;
        const int lc = (((pch >> 2) ^ (r & 3)) << 2) | ((pch & 3) ^ (((r >> 3) & 1) << 1));     // logical chunk of this physical one (swz_cw)
        if (lc < T.zchunks) {
            zp[jj] = reinterpret_cast<const char*>(T.Z + T.n0 + lc * 8) + (int64_t)(s0 * 32 + r) * ldz2;
            zinc[jj] = (unsigned)(32 * ldz2); zr[jj] = r;
        } else { zp[jj] = zpage; zinc[jj] = 0u; zr[jj] = -1; }
    }
    const unsigned my_x = (unsigned)__builtin_amdgcn_readfirstlane(NX * lw) * 1024u, my_z = XB + (unsigned)__builtin_amdgcn_readfirstlane(4 * lw) * 1024u;
    int sl = s0;                                                     // slab the next issue fetches
    auto issue = [&](int slot) __attribute__((always_inline)) {
        const unsigned base = lds0 + (unsigned)slot * SLAB;
        // fast: every row of the slab and its halo lies inside the batch (the halo row of slab 0 would be row -1)
        const bool fast = sl >= 1 && 32 * sl + 33 <= m_rows;
        if (fast) {
#pragma unroll
            for (int jj = 0; jj < NX; ++jj) { if (!(CW_ABL & 1)) dma16(xp[jj], base + my_x + 1024u * jj); xp[jj] += xinc[jj]; }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) { if (!(CW_ABL & 1)) dma16(zp[jj], base + my_z + 1024u * jj); zp[jj] += zinc[jj]; }
        } else {
#pragma unroll
            for (int jj = 0; jj < NX; ++jj) {
                const int row = 32 * sl + xr[jj] - HALO;
                dma16((xr[jj] >= 0 && (row < 0 || row >= m_rows)) ? zpage : xp[jj], base + my_x + 1024u * jj);
                xp[jj] += xinc[jj];
            }
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                dma16((zr[jj] >= 0 && 32 * sl + zr[jj] >= m_rows) ? zpage : zp[jj], base + my_z + 1024u * jj);
                zp[jj] += zinc[jj];
            }
        }
        ++sl;
    };
    int nxt = -1;
    if (fetch) asm volatile("global_atomic_add %0, %1, %2, off sc0" : "+v"(nxt) : "v"(pa.counters + q), "v"(1) : "memory");
#pragma unroll
    for (int i_ = 0; i_ < SLOTS - 1; ++i_) issue(i_);                // slabs 0 .. SLOTS - 2
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"((SLOTS - 2) * NI) : "memory");          // slab 0 has landed
    if (fetch) {
        asm volatile("" : "+v"(nxt));
        if (nxt < 0) asm volatile("s_waitcnt vmcnt(0)" : "+v"(nxt) :: "memory");
        next_slot[par ^ 1] = nxt;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int isl = SLOTS - 1;                                             // slab s + SLOTS - 1 goes where slab s - 1 was
    for (int s = 0; s < nsl; ++s) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"((SLOTS - 3) * NI) : "memory");      // slab s + 1 has landed
        CW3_BAR();                                                                   // ... and slab s - 1 is out of use
        issue(isl);
        isl = isl + 1 == SLOTS ? 0 : isl + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the run-ahead pieces must not land in the next entry's slabs
    __builtin_amdgcn_s_barrier();                                    // the compute waves have read the last slab: the ring is free for the next entry
}

// ---- compute wave (wm, wn): group slots 4 wm .. 4 wm + 3 x columns 112 wn .. 112 wn + 111 of the tile.
template <int W>
__device__ __forceinline__ void cw3_compute(const Cw3Args& pa, const Cw3Tile* __restrict__ Tg, const Cw3Tile& T, int s0, int nsl, int wid, int lane, int tid,
                                            u16* cw_ring, unsigned lds0, int& dslot) {
    constexpr int IT = 4;
    constexpr int SLOTS = Cw3Geo<W>::SLOTS;
    constexpr unsigned XB = Cw3Geo<W>::XB, SLAB = Cw3Geo<W>::SLAB;
    const int wm = (wid >> 1) & 3, wn = wid & 1;
    typedef s16x4_t __attribute__((address_space(3))) * lds_v4;
    typedef char __attribute__((address_space(3))) * lds_c;
    const lds_c ring_c = (lds_c)cw_ring;         // (LDS addresses below are byte offsets from the ring's base)
    // per slot: tap (row offset of the read), X-tile window, channel of the flush
    int tap[IT], cch[IT];
    int fo0[IT], fo1[IT];                       // byte offsets of the two transposing reads inside a slab (rows r0 + shift, r0 + 4 + shift)
    const int r0 = 8 * (lane >> 4) + ((lane & 15) >> 2);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int sidx = 4 * wm + i;
        tap[i] = __builtin_amdgcn_readfirstlane((int)Tg->slot_tap[sidx]);
        const int win = __builtin_amdgcn_readfirstlane((int)Tg->slot_win[sidx]);
        cch[i] = tap[i] >= 0 ? __builtin_amdgcn_readfirstlane((int)Tg->win_c[win]) : -1;
        const int sh = (W == 6 && tap[i] >= 0) ? tap[i] : 0;
        fo0[i] = cw3_xoff<W>(r0 + sh, win & 15) + (lane & 3) * 8;
        fo1[i] = cw3_xoff<W>(r0 + 4 + sh, win & 15) + (lane & 3) * 8;
    }
    // dZ fragments: byte offset of (row r0, column 112 wn + 16 j + 4 (lane & 3)) = r0 * 512 + 8 (lane & 3) + (2 col_j ^ 2 P), P = the row's
    // swizzle bits (swz_cw): col_j touches bits 5..8 only, so ONE per-lane register and a scalar per column group address all seven
    const unsigned az0 = XB + (unsigned)(r0 * 512 + (lane & 3) * 8 + 2 * (((r0 & 3) << 5) | (((r0 >> 3) & 1) << 4)));
    unsigned kz[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) kz[j] = (unsigned)__builtin_amdgcn_readfirstlane(2 * (wn * 112 + j * 16));
    f32x4_t acc[IT][7];
#pragma unroll
    for (int i = 0; i < IT; ++i)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // level of the slab's first row, and from it the slab-local row that is level 0 (`bnd`; the row before it is level seq - 1):
    // tap 0 must not see row bnd (its shifted source is the previous column's last level), tap 2 must not see row bnd - 1
    int lv0 = (int)(((int64_t)s0 * 32) % pa.seq);
    auto next_bnd = [&]() __attribute__((always_inline)) { const int b = lv0 == 0 ? 0 : pa.seq - lv0; lv0 += 32; if (lv0 >= pa.seq) lv0 -= pa.seq; return b; };
    const unsigned zaddr = CW3_ZERO_OFF;
    // Column boundaries.  Per slot a per-lane KEY: the slab-local row whose being level 0 makes this lane's FIRST read invalid - r0 under
    // tap 0 (its source is the previous column's last level), r0 + 1 under tap 2 (the row itself is the last level), never under tap 1;
    // the second read's key is 4 more.  A slab holds at most one level-0 row `bnd` (seq >= 34): one compare + one select per read.
    // Empty slots read like any other: their sums are never flushed.
    int key[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) key[i] = (W == 6 && tap[i] == 0 && !(CW3_EXP & 16)) ? r0 : ((W == 6 && tap[i] == 2 && !(CW3_EXP & 16)) ? r0 + 1 : -1000 - (lane & 1));
    auto read_h = [&](bf16x8_t& dst, int i, unsigned slab, int bnd) __attribute__((always_inline)) {
        union { bf16x8_t v; s16x4_t h[2]; } u_;
        unsigned a0 = slab + (unsigned)fo0[i], a1 = slab + (unsigned)fo1[i];
        if (W == 6 && !(CW3_EXP & 2)) {                              // (keys are <= 28: a slab whose level-0 row lies beyond 32 matches nothing)
            if (key[i] == bnd) a0 = zaddr;
            if (key[i] == bnd - 4) a1 = zaddr;
        }
        u_.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(ring_c + a0));
        u_.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(ring_c + a1));
        dst = u_.v;
    };
    auto read_z = [&](bf16x8_t& dst, int j, unsigned az) __attribute__((always_inline)) {     // az = az0 + the slab's offset
        union { bf16x8_t v; s16x4_t h[2]; } u_;
        const unsigned a = az ^ kz[j];
        u_.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(ring_c + a));
        u_.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(ring_c + a + 2048u));
        dst = u_.v;
    };
    // Pipeline.  fh[i] (the wave's four group slots) stay in registers for a whole slab; the seven dZ column groups stream through FOUR
    // fragment buffers: b0 {g0, g4}, b1 {g1, g5}, b2 {g2, g6}, b3 {g3}, each refilled right behind the MFMAs that used it - from the slab
    // in use, or from the next one (which the barrier at the top of the slab promised has landed) - so that every slab starts with g0 ..
    // g3 in b0 .. b3 and a request is at least three groups (12 MFMAs) old when it is used.  The last two groups run slot-major so
    // that fh[i] of the next slab is requested behind the slot's last MFMA, 6 .. 0 MFMAs before the slab ends and i more before its use.
    bf16x8_t fh[IT], fz[4];
    __builtin_amdgcn_s_barrier();                                    // slab 0 has landed
    cw3_stamp(pa, tid, dslot, W);                                    // (tag: the tile type, for tools/cnn_wgrad_stamps.py)
    int bnd_next;
    {
        const int bnd = next_bnd();
#pragma unroll
        for (int i = 0; i < IT; ++i) read_h(fh[i], i, 0u, bnd);
#pragma unroll
        for (int j = 0; j < 4; ++j) read_z(fz[j], j, az0);
        bnd_next = next_bnd();
    }
    unsigned cur_off = 0, nx_off = (1 % SLOTS) * SLAB;
#define CW3_GROUP(j, b)                                                                                                      \
    _Pragma("unroll") for (int i = 0; i < IT; ++i)                                                                           \
        if (!(CW_ABL & 4)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i], fz[b], acc[i][j], 0, 0, 0);
    for (int s = 0; s < nsl; ++s) {
        CW3_BAR();                                                   // slab s + 1 has landed, slab s - 1 is free
        const unsigned azc = az0 + cur_off, azn = az0 + nx_off, xn = nx_off;
        CW3_GROUP(0, 0) if (!(CW_ABL & 2)) read_z(fz[0], 4, azc); CW3_SB();
        CW3_GROUP(1, 1) if (!(CW_ABL & 2)) read_z(fz[1], 5, azc); CW3_SB();
        CW3_GROUP(2, 2) if (!(CW_ABL & 2)) read_z(fz[2], 6, azc); CW3_SB();
        CW3_GROUP(3, 3) if (!(CW_ABL & 2)) read_z(fz[3], 3, azn); CW3_SB();
        CW3_GROUP(4, 0) if (!(CW_ABL & 2)) read_z(fz[0], 0, azn); CW3_SB();
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            if (!(CW_ABL & 4)) {
                acc[i][5] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i], fz[1], acc[i][5], 0, 0, 0);
                acc[i][6] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fh[i], fz[2], acc[i][6], 0, 0, 0);
            }
            if (!(CW_ABL & 2)) read_h(fh[i], i, xn, bnd_next);
            CW3_SB();
        }
        if (!(CW_ABL & 2)) { read_z(fz[1], 1, azn); read_z(fz[2], 2, azn); }
        bnd_next = next_bnd();
        cur_off = nx_off;
        nx_off = nx_off + SLAB == SLOTS * SLAB ? 0u : nx_off + SLAB;
    }
#undef CW3_GROUP
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                    // done with the ring (the loaders refill it with the next entry's slabs during the flush)
    cw3_stamp(pa, tid, dslot, 0);
    // ---- flush (conv_wgrad2.h): through LDS, eight rows at a time, 64 consecutive floats of one row per atomic instruction
    if (!(CW_ABL & 8)) {
        float* stg = reinterpret_cast<float*>(reinterpret_cast<char*>(cw_ring) + CW3_STAGE_OFF) + wid * (8 * 112);
        const int nmax = T.cout - (T.n0 + wn * 112);                 // columns of this wave that exist
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            if (cch[i] >= 0) {
                float* row = T.dW + ((int64_t)tap[i] * T.cin + cch[i]) * T.cout + T.n0 + wn * 112;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {                     // rows 8 hh .. 8 hh + 7 of the group live in lanes 32 hh .. 32 hh + 31
                    if ((lane >> 5) == hh) {
#pragma unroll
                        for (int j = 0; j < 7; ++j)
#pragma unroll
                            for (int r = 0; r < 4; ++r) stg[(4 * ((lane >> 4) & 1) + r) * 112 + j * 16 + (lane & 15)] = acc[i][j][r];
                    }
                    for (int rr = 0; rr < 8 && cch[i] + 8 * hh + rr < T.cin; ++rr) {
                        float* dst = row + (int64_t)(8 * hh + rr) * T.cout;
                        const float v0 = stg[rr * 112 + lane];
                        if (lane < nmax) cw3_atomic_add(dst + lane, v0);
                        if (lane < 48) {
                            const float v1 = stg[rr * 112 + 64 + lane];
                            if (64 + lane < nmax) cw3_atomic_add(dst + 64 + lane, v1);
                        }
                    }
                }
            } else if (cch[i] == -2 && T.db && (lane >> 4) == 0) {   // the ones row: bias gradient
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    const int n = T.n0 + wn * 112 + j * 16 + (lane & 15);
                    if (n < T.cout) cw3_atomic_add(T.db + n, acc[i][j][0]);
                }
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < IT; ++i)
#pragma unroll
            for (int j = 0; j < 7; ++j) asm volatile("" :: "v"(acc[i][j]));
    }
    cw3_stamp(pa, tid, dslot, 0);
}

__global__ __launch_bounds__(512 + 64 * CW2L_LOADERS) void k_conv_wgrad3l(const Cw3Args pa) {
    extern __shared__ __attribute__((aligned(16))) u16 cw_ring[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef u16 __attribute__((address_space(3))) * lds_p;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_p)cw_ring);
    int* next_slot = reinterpret_cast<int*>(reinterpret_cast<char*>(cw_ring) + CW3_QWORD_OFF);   // [2]: entry k + 1 of the queue, by parity
    const int q = blockIdx.x & 7;
    const int qb = pa.q_begin[q], qn = pa.q_begin[q + 1] - qb;
    int dslot = 2;
    cw3_stamp_head(pa, tid);
    if (tid < 4) reinterpret_cast<int*>(reinterpret_cast<char*>(cw_ring) + CW3_ZERO_OFF)[tid] = 0;
    int k;
    if (pa.counters) {
        if (tid == 0) next_slot[0] = atomicAdd(pa.counters + q, 1);
        __syncthreads();
        k = next_slot[0];
    } else {
        __syncthreads();
        k = blockIdx.x >> 3;
    }
    for (int par = 0; k < qn; par ^= 1) {
        const CwWork wk_ = pa.work[qb + k];
        const int tile_u = __builtin_amdgcn_readfirstlane(wk_.tile);
        const int s0 = __builtin_amdgcn_readfirstlane(wk_.s0);
        const int nsl = __builtin_amdgcn_readfirstlane(wk_.s1) - s0;     // >= 1 (the host writes no empty entries)
        const Cw3Tile* Tg = pa.tiles + tile_u;
        const Cw3Tile T = *Tg;
        cw3_stamp(pa, tid, dslot, nsl);
        if (wid >= 8) {
            const bool fetch = pa.counters && wid == 8 && lane == 0;
            if (T.ntap == 3) cw3_loader<6, 34, 2>(pa, Tg, T, s0, nsl, wid - 8, lane, lds0, q, par, next_slot, fetch);
            else cw3_loader<16, 32, 4>(pa, Tg, T, s0, nsl, wid - 8, lane, lds0, q, par, next_slot, fetch);
        } else {
            if (T.ntap == 3) cw3_compute<6>(pa, Tg, T, s0, nsl, wid, lane, tid, cw_ring, lds0, dslot);
            else cw3_compute<16>(pa, Tg, T, s0, nsl, wid, lane, tid, cw_ring, lds0, dslot);
        }
        if (!pa.counters) break;
        k = next_slot[par ^ 1];
    }
    if (pa.dbg && tid == 0 && dslot < CW_DBG_SLOTS) pa.dbg[(int64_t)blockIdx.x * CW_DBG_SLOTS + dslot] = __builtin_amdgcn_s_memrealtime() | (1ull << 63);
}
