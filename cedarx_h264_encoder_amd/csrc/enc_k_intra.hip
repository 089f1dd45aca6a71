// cedarx_h264_encoder_amd/csrc/enc_k_intra.hip
// k_intra_rows: closed-loop intra prediction + transform + reconstruction of an I picture as a wavefront of 4x4 BLOCKS.
//
// Intra4x4 prediction of block (X, Y) -- 4x4-block coordinates of the picture -- reads the reconstructed blocks (X-1, Y),
// (X-1, Y-1), (X, Y-1) and (X+1, Y-1): with T = X + 2Y every block of a picture can be coded at time T, 4*mb_w + 2*(4*mb_h - 1)
// block times for the picture (1080p: 1022) whatever a kernel does. k_intra_recon2 (enc_k_wavefront.hip) steps in whole
// macroblocks -- ten dependent block times before the next macroblock of the row may start: 254 x 10 block times -- and
// its block time holds three trips through LDS. Here:
//   * ONE WAVE PER MACROBLOCK ROW. Its four lane groups are the four block rows of that macroblock row, group r two blocks
//     behind group r-1; the 16 lanes of a group are the 16 samples of the group's current block. A step of the wave is
//     one block time for all four rows.
//   * THE CHAIN STAYS IN REGISTERS. The right column of the block just reconstructed (the next block's left neighbours) is
//     gathered inside the lane group with DPP; its bottom row is packed into lane 15 of the group and reaches the group
//     below with ONE v_mov_b32_dpp row_bcast:15; the rows above that are two and three steps old are that same register,
//     kept. The edge samples a lane needs sit in four dwords (left column, the three dwords above), and EVERY predictor
//     of 8.3.1.2 -- the nine modes, DC with its four availability cases included -- is
//         ( sum of 4 bytes picked from {above, above-right} + sum of 4 bytes picked from {left, above-left} + rnd ) >> sh
//     i.e. two v_perm_b32, two v_sad_u8 and a shift, with the byte selectors, rnd and sh read from a table in LDS by
//     (predictor, lane) one step ahead: (a + 2b + c + 2) >> 2 picks a, b, b, c; (b + c + 1) >> 1 picks b, c, b, c; a copy
//     picks b four times; DC picks the four samples above and the four to the left.
//   * lanes sit in boustrophedon order (sample row i, position j or 3 - j for odd i), so that all four butterfly partners
//     of the 4x4 transforms are single DPP operands: quad_perm for the horizontal pairs, row_mirror (rows 0-3, 1-2) and
//     row_half_mirror (rows 0-1, 2-3) for the vertical ones.
//   * source samples come from an LDS ring a stager wave fills, reconstruction and levels go to LDS rings a writer wave
//     empties; neither is on the chain. Rows inside a band of 4 macroblock rows (one workgroup, 4 chain waves on 4 SIMDs)
//     meet through full-row buffers of {4 samples, valid} entries in LDS: the payload says itself that it has arrived.
//     Bands hand their last row on through memory as tagged 8-byte granules (writer -> courier), as k_deblock_rows does.
//   * an Intra16x16 or I_PCM macroblock drains the skew (its prediction needs the whole left neighbour), runs with the
//     16 lanes of group 0 as its 16 blocks on a tile in LDS, and the skew builds up again behind it.
//   * chroma prediction only reads chroma: the chroma planes are workgroups of their own in the same launch, a wave per
//     macroblock row, rows one macroblock apart, the 4 lane groups = the 4 blocks of a plane.
// Bytes are those of k_intra_recon2 (tests/test_gpu_parity.py runs both; CEDAR_INTRA_ROWS=0 keeps the old kernel).
//
// The reference delegates this stage to silicon (kernel/cedar.c:1155-1163 selects the I slice; :1135-1153 programs the
// reconstruction addresses; :1176 triggers).
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "enc_device.h"
#include "enc_wf_common.h"

namespace cedar {

constexpr int IRW_WAVES = 7;          // 4 chain waves (one per macroblock row and SIMD), stager, writer, courier
constexpr int IRW_RING = 8;           // macroblocks of source / reconstruction / levels a row keeps in LDS (power of two)
constexpr int IRW_RB = IRW_RING * 16; // bytes of a ring row
constexpr int IRW_XB = IRW_RING * 4;  // 4x4-block columns of a ring
constexpr int IRW_TAB_ROWS = 22;      // predictors: 9 modes with / without the samples above-right, DC by availability
constexpr int IRW_T16 = 32;           // stride of the Intra16x16 tile: sample (x, y) at [(1 + y) * 32 + 8 + x]
constexpr int IRW_TC = 32;            // stride of the chroma tile (nv12): sample (x, y, c) at [(1 + y) * 32 + 8 + 2x + c]

struct IrwTabEnt {
	uint32_t sel_a; // v_perm_b32 selector over { above-left dword (bytes 4-7), left column (bytes 0-3) }
	uint32_t sel_b; // ... over { above-right dword (4-7), above dword (0-3) }
	uint32_t rnd, sh;
};

// one macroblock row of a luma band
struct __attribute__((aligned(128))) IrwRowY {
	uint8_t io[16][IRW_RB];        // staged source, sample row y, column x & 127; the reconstruction takes its place block by block
	int16_t lev[4][IRW_XB][16];    // levels: block row, block column X & 31, zig-zag index
	int16_t dc16[IRW_RING][16];    // Intra16x16 DC levels, zig-zag
	uint16_t ctl[4][IRW_XB];       // per 4x4 block: byte offset of its predictor's row in the table
	uint32_t info[IRW_RING];       // first dword of the macroblock's record (type, Intra16x16 mode, chroma mode)
	uint8_t t16[17 * IRW_T16];     // Intra16x16: the macroblock with the row above it and the column to its left
	int dc[16], dcl[16];
};

// one macroblock row of a chroma band
struct __attribute__((aligned(128))) IrwRowC {
	uint8_t io[8][IRW_RB];         // staged source (nv12 rows of 8 CbCr pairs), ring by macroblock; the reconstruction takes its place
	int16_t lev[IRW_RING][8][16];  // AC levels: macroblock, (plane, block), zig-zag index (entry 0 unused)
	int16_t cdc[IRW_RING][8];      // DC levels: (plane, index)
	uint8_t nnz[IRW_RING][8];
	uint32_t info[IRW_RING];
	uint32_t acdc[IRW_RING];       // the chroma bits of the coded block pattern: 0x20 an AC level, else 0x10 a DC level
	uint8_t t[9 * IRW_TC];         // the macroblock with the row above it and the column to its left (persists along the row)
	int dcw[8], dcl[8];
};

template <int PL> struct IrwShared;
template <> struct __attribute__((aligned(128))) IrwShared<1> {
	IrwRowY g[4];
	IrwTabEnt tab[IRW_TAB_ROWS][16];
	int staged[4];  // macroblocks of row g staged
	int written[4]; // macroblocks of row g written back (their ring slots are free)
	int ticket, abort;
};
template <> struct __attribute__((aligned(128))) IrwShared<2> {
	IrwRowC g[4];
	int staged[4];
	int written[4];
	int ticket, abort;
};

extern __shared__ __attribute__((aligned(128))) uint8_t irw_lds[];

// (every LDS operation of a wave is executed in order and the CU has one LDS: a compiler barrier is all the release /
// acquire a payload-then-flag hand-off between two waves of a workgroup needs -- no s_waitcnt lgkmcnt(0) on the chain)
#define IRW_ORDER() asm volatile("" ::: "memory")

#define IRW_DPP(v, ctrl) __builtin_amdgcn_update_dpp(0, (int)(v), (ctrl), 0xf, 0xf, true)

// A rowbuf entry: { 4 samples, 1 }. Written once per kernel by one ds_write_b64, polled with ds_read_b64.
DEVI uint64_t irw_entry_load(const uint64_t *e) { return __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
DEVI void irw_entry_store(uint64_t *e, uint32_t samples)
{
	__hip_atomic_store(e, (uint64_t)samples | 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct IrwWait {
	WaitCtx w;
	bool dead; // a wait of this wave ran out (or the workgroup was told to stop): no later wait waits
};

// wave-uniform bounded spin: `ready` is polled with a short sleep, then with a long one and a look at the clock
template <typename F>
DEVI bool irw_spin(F ready, IrwWait &x)
{
	if (x.dead)
		return false;
	uint64_t t0 = 0;
	for (int spins = 0; !ready(); spins++) {
		if (spins < 1024) {
			__builtin_amdgcn_s_sleep(1);
			continue;
		}
		if (wait_expired((uint32_t)spins, t0)) {
			__hip_atomic_store(x.w.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(x.w.abort_lds, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			xstore32(x.w.abort_x, 1u);
			x.dead = true;
			return false;
		}
		if (__builtin_amdgcn_readfirstlane(__hip_atomic_load(x.w.abort_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))) {
			x.dead = true;
			return false;
		}
		__builtin_amdgcn_s_sleep(32);
	}
	IRW_ORDER();
	return true;
}

DEVI int irw_ctr(const int *word) { return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// ... read by a whole wave for a wave-uniform decision (the compiler cannot know that every lane loads the same word)
DEVI int irw_ctr_u(const int *word) { return __builtin_amdgcn_readfirstlane(irw_ctr(word)); }
// a lane constant the compiler must not see through: +-1 would turn a v_mul_i32_i24 into compare / negate / select
DEVI int irw_opaque(int v)
{
	asm volatile("" : "+v"(v));
	return v;
}

// forward / inverse 4x4 core transform, one sample per lane in boustrophedon order: lane 4i + p holds (row i, column j),
// j = p for even i, 3 - p for odd i. Every butterfly partner is one DPP operand. ci / cj: lane_consts(i), lane_consts(j).
// fdct returns W[P[i]][P[j]], P = {0, 2, 1, 3}; idct takes the scaled coefficient (P[i], P[j]) -- the rounding 32 already
// added to the DC coefficient -- and returns the residual sample (rows first, then columns: 8.5.12.2).
DEVI int irw_fdct(int v, const LaneK &ci, const LaneK &cj)
{
	int s = IRW_DPP(v, 0x1B) + __mul24(cj.s1, v);                 // j <-> 3 - j: a b c d
	s = __mul24(cj.fa, s) + __mul24(cj.fb, IRW_DPP(s, 0xB1));     // j <-> j ^ 1: W0 W2 W1 W3
	s = IRW_DPP(s, 0x140) + __mul24(ci.s1, s);                    // row_mirror: i <-> 3 - i
	// (opaque: the quantiser's 24-bit multiply only reads the low 24 bits, and the compiler would pick a full-width
	// multiply-add for a value of which only those are wanted)
	return irw_opaque(__mul24(ci.fa, s) + __mul24(ci.fb, IRW_DPP(s, 0x141))); // row_half_mirror: i <-> i ^ 1
}

DEVI int irw_idct(int d, const LaneK &ci, const LaneK &cj)
{
	int e = __mul24(cj.ia, d >> cj.sh) + __mul24(cj.ib, IRW_DPP(d, 0xB1));
	int f = IRW_DPP(e, 0x1B) + __mul24(cj.s1, e);
	e = __mul24(ci.ia, f >> ci.sh) + __mul24(ci.ib, IRW_DPP(f, 0x141));
	f = IRW_DPP(e, 0x140) + __mul24(ci.s1, e);
	return f >> 6;
}

// sign(w) * ((|w| * mf + f) >> qbits), clipped to +-LEVEL_MAX, on the signed product: c2 = (1 << qbits) - 1 - 2 f
DEVI int irw_quant(int w, int mf, int f, int c2, int qbits)
{
	const int pr = __mul24(w, mf);
	const int lv = (pr + f + ((pr >> 31) & c2)) >> qbits;
	return min(max(lv, -LEVEL_MAX), LEVEL_MAX);
}

// lane constants of a sample lane (ql = lane & 15)
struct IrwLane {
	int qi, qj;      // sample row, sample column
	LaneK ci, cj;
	int pclass, zz;  // position class and zig-zag index of the coefficient the lane holds after the forward transform
	int dc_lane;     // holds coefficient (0, 0)
};

DEVI IrwLane irw_lane(int ql)
{
	IrwLane L;
	const int qp = ql & 3;
	L.qi = ql >> 2;
	L.qj = (L.qi & 1) ? 3 - qp : qp;
	L.ci = lane_consts(L.qi);
	L.cj = lane_consts(L.qj);
	for (LaneK *k : { &L.ci, &L.cj }) {
		k->s1 = irw_opaque(k->s1);
		k->fa = irw_opaque(k->fa);
		k->fb = irw_opaque(k->fb);
		k->ia = irw_opaque(k->ia);
		k->ib = irw_opaque(k->ib);
	}
	const int kv = L.qi == 1 ? 2 : L.qi == 2 ? 1 : L.qi, kh = L.qj == 1 ? 2 : L.qj == 2 ? 1 : L.qj;
	const int rpos = 4 * kv + kh;
	L.pclass = kPosClass[rpos];
	int zz = 0;
#pragma unroll
	for (int z = 0; z < 16; z++)
		zz = kZigzag[z] == rpos ? z : zz;
	L.zz = zz;
	L.dc_lane = rpos == 0;
	return L;
}

// ------------------------------------------------------------------------------------------------------------------
// the predictor table: entry [row][lane] for the lane's sample (i, j). Rows 0..8: Intra4x4 mode with the samples
// above-right, 9..17: without them (E[10..13] fold onto E[9]); 18..21: DC with nothing / above / left / both.
// Edge array of 8.3.1.2 as pred4_px (enc_device.h) reads it: E[1..4] = left column bottom-up, E[5] = above-left,
// E[6..13] = the row above. Bytes: left column dword L (byte i = row i), P2 = dword above-left (its byte 3), P1 = dword
// above, A = dword above-right.
// ------------------------------------------------------------------------------------------------------------------
DEVI void irw_edge_sel(int k, bool tr, int slot, uint32_t &sa, uint32_t &sb)
{
	k = k < 1 ? 1 : k > 13 ? 13 : k;
	if (!tr && k >= 10)
		k = 9;
	uint32_t a = 0x0cu, b = 0x0cu; // 0x0c: the constant 0
	if (k <= 4)
		a = (uint32_t)(4 - k);      // L: E[4] = row 0 ... E[1] = row 3
	else if (k == 5)
		a = 7u;                     // byte 3 of P2
	else
		b = (uint32_t)(k - 6);      // P1 bytes 0-3, A bytes 4-7
	sa |= a << (8 * slot);
	sb |= b << (8 * slot);
}

DEVI IrwTabEnt irw_tab_entry(const uint8_t *pred4_tab, int row, int qi, int qj)
{
	IrwTabEnt t{ 0x0c0c0c0cu, 0x0c0c0c0cu, 128u, 0u };
	if (row >= 18) {
		const bool top = (row - 18) & 1, left = (row - 18) & 2;
		if (left)
			t.sel_a = 0x03020100u;
		if (top)
			t.sel_b = 0x03020100u;
		t.rnd = top && left ? 4u : (top || left) ? 2u : 128u;
		t.sh = top && left ? 3u : (top || left) ? 2u : 0u;
		return t;
	}
	const bool tr = row < 9;
	const int mode = tr ? row : row - 9;
	const int ent = pred4_tab[mode * 16 + 4 * qi + qj];
	const int kind = ent >> 4, k = ent & 15;
	if (kind == 3) // (mode 2 never comes through these rows)
		return t;
	const int ka = k > 0 ? k - 1 : 0, kc = k + 1;
	const int s0 = kind == 2 ? ka : k, s1 = kind == 1 ? kc : k, s2 = k, s3 = kind == 0 ? k : kc;
	t.sel_a = t.sel_b = 0;
	irw_edge_sel(s0, tr, 0, t.sel_a, t.sel_b);
	irw_edge_sel(s1, tr, 1, t.sel_a, t.sel_b);
	irw_edge_sel(s2, tr, 2, t.sel_a, t.sel_b);
	irw_edge_sel(s3, tr, 3, t.sel_a, t.sel_b);
	t.rnd = 2u;
	t.sh = 2u;
	return t;
}

// 4x4 Hadamard sign H[i][a] for rows (1,1,1,1) (1,1,-1,-1) (1,-1,-1,1) (1,-1,1,-1)
DEVI int irw_hsign(int i, int a)
{
	const uint32_t neg = (0xAu << 12) | (0x6u << 8) | (0xCu << 4) | 0x0u;
	return ((neg >> (4 * i + a)) & 1) ? -1 : 1;
}

// coding index of the 4x4 block at (bx, by)
DEVI int irw_blk_idx(int bx, int by) { return ((by >> 1) << 3) | ((bx >> 1) << 2) | ((by & 1) << 1) | (bx & 1); }

template <int PL>
DEVI void intra_rows_body(const Params &p)
{
	constexpr bool LUMA = PL == 1;
	IrwShared<PL> &S = *reinterpret_cast<IrwShared<PL> *>(irw_lds);
	const int nbx = 4 * p.mb_w;          // entries of a row: 4x4-block columns (luma), dwords of 2 CbCr pairs (chroma)
	const int RS = nbx + 2;              // ... with one in front (index X + 1) and one behind
	uint64_t *const rows = reinterpret_cast<uint64_t *>(irw_lds + sizeof(IrwShared<PL>)); // [5][RS]: 0 the band above, 1..4 the band's rows
	const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // (uniform, and the compiler should know)
	const int NB = p.ir_bands;
	uint32_t *sync = p.wf_sync; // region 0
	if (threadIdx.x == 0) {
		S.ticket = (int)atomicAdd(&sync[LUMA ? 0 : 2], 1u); // bands are taken in the order workgroups start: whoever waits, waits for a resident one
		S.abort = 0;
	}
	if (threadIdx.x < 4) {
		S.staged[threadIdx.x] = 0;
		S.written[threadIdx.x] = 0;
	}
	for (int i = threadIdx.x; i < 5 * RS; i += blockDim.x)
		rows[i] = 0;
	if constexpr (LUMA) {
		for (int i = threadIdx.x; i < IRW_TAB_ROWS * 16; i += blockDim.x) {
			const IrwLane L = irw_lane(i & 15);
			S.tab[i >> 4][i & 15] = irw_tab_entry(p.pred4_tab, i >> 4, L.qi, L.qj);
		}
		for (int i = threadIdx.x; i < 4 * 4 * IRW_XB; i += blockDim.x)
			S.g[i / (4 * IRW_XB)].ctl[(i / IRW_XB) & 3][i % IRW_XB] = 0;
	}
	__syncthreads();
	// a ticket beyond the launch means the sync words were not zero when the kernel started: refuse
	// to index memory with it (code 32 in the frame's error word) rather than fault
	if (__builtin_amdgcn_readfirstlane(S.ticket) < 0 || __builtin_amdgcn_readfirstlane(S.ticket) >= p.n_streams * p.slice_num * NB) {
		if (threadIdx.x == 0)
			__hip_atomic_store(p.wf_err, 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		return;
	}
	const int ticket = __builtin_amdgcn_readfirstlane(S.ticket);
	const int unit = ticket / NB, band = ticket % NB;
	const int st = unit / p.slice_num, sl = p.slice_first + unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, sl, &r0, &r1);
	const int nbands = (r1 - r0 + 3) >> 2;
	if (band >= nbands) // (slices of a picture differ by one row: the launch has the bands of the tallest)
		return;
	if (threadIdx.x == 0)
		dbg_event(p, 3, p.db_tag, (uint32_t)band | (uint32_t)PL << 8 | (uint32_t)unit << 16);
	IrwWait xw{ WaitCtx{ p.wf_err, &S.abort, sync + 1 }, false };
	const int y0b = r0 + 4 * band;
	const int nrows = min(4, r1 - y0b);
	const bool feeds_next = band + 1 < nbands; // the band's last row goes to the band below through memory
	const bool from_above = band > 0;
	// granules per unit, plane and band: one per rowbuf entry, { 4 samples, the frame's tag }
	uint64_t *hb_out = p.ir_hbuf + (((size_t)unit * 2 + (LUMA ? 0 : 1)) * NB + band) * nbx;
	const uint64_t *hb_in = p.ir_hbuf + (((size_t)unit * 2 + (LUMA ? 0 : 1)) * NB + (band > 0 ? band - 1 : 0)) * nbx;
	auto entry = [&](int row, int X) -> uint64_t * { return rows + row * RS + X + 1; };

	if (wave < 4) {
		// ======================================== chain wave of macroblock row `wave` ========================================
		const int gr = wave;
		const int my = y0b + gr;
		if (my >= r1)
			return;
		__builtin_amdgcn_s_setprio(3);
		const bool top_mb = my > r0;
		const int g = lane >> 4, ql = lane & 15;
		const IrwLane LN = irw_lane(ql);
		const int qi = LN.qi, qj = LN.qj;
		const LaneK ci = LN.ci, cj = LN.cj;
		auto wait_staged = [&](int need) {
			irw_spin([&]() { return irw_ctr_u(&S.staged[gr]) >= need; }, xw);
		};
		auto wait_entry = [&](int row, int X) -> uint32_t { // the samples of an entry, once it is there
			uint64_t v = irw_entry_load(entry(row, X));
			if (!(uint32_t)__builtin_amdgcn_readfirstlane((int)(v >> 32))) {
				irw_spin([&]() { v = irw_entry_load(entry(row, X)); return __builtin_amdgcn_readfirstlane((int)(v >> 32)) != 0; }, xw);
			}
			return (uint32_t)v;
		};

		if constexpr (LUMA) {
			IrwRowY &R = S.g[gr];
			const int pclass = LN.pclass, zz = LN.zz;
			const int my_mf = pclass == 0 ? p.mf[0] : pclass == 1 ? p.mf[1] : p.mf[2];
			const int my_dv = pclass == 0 ? p.dv[0] : pclass == 1 ? p.dv[1] : p.dv[2];
			const int dvq = my_dv << p.qshift, rnd32 = LN.dc_lane ? 32 : 0;
			const int qbits = p.qbits, f_intra = (1 << qbits) / 3, c2 = (1 << qbits) - 1 - 2 * f_intra;
			const uint32_t sh_row = 8u * (uint32_t)qi, sh_col = 8u * (uint32_t)qj;
			const uint8_t *const tabl = reinterpret_cast<const uint8_t *>(&S.tab[0][ql]);
			uint8_t *const io_l = &R.io[4 * g + qi][0]; // (a sample is fetched a step before its reconstruction is written over it)
			int16_t *const lev_l = &R.lev[g][0][zz];
			const uint16_t *const ctl_l = &R.ctl[g][0];

			// the bottom row of the group's block, one byte per sample column, in lane 15 of the group
			auto rowpack = [&](int rec) -> uint32_t {
				uint32_t y = (uint32_t)rec << sh_col;
				y |= (uint32_t)IRW_DPP(y, 0x111); // row_shr:1
				y |= (uint32_t)IRW_DPP(y, 0x112); // row_shr:2
				return y;
			};
			// the right column of the group's block, byte i = sample row i, in every lane of the group
			auto colpack = [&](int rec) -> uint32_t {
				int q = __builtin_amdgcn_update_dpp(0, rec, 0xFF, 0xf, 0x5, false); // even rows: column 3 sits at position 3
				q = __builtin_amdgcn_update_dpp(q, rec, 0x00, 0xf, 0xA, false);     // odd rows: at position 0
				uint32_t z = (uint32_t)q << sh_row;
				z |= (uint32_t)IRW_DPP(z, 0x124); // row_ror:4
				z |= (uint32_t)IRW_DPP(z, 0x128); // row_ror:8
				return z;
			};

			int rec = 0;
			uint32_t y3 = 0, P1 = 0, P2 = 0;
			int mx = 0;
			while (mx < p.mb_w && !xw.dead) {
				wait_staged(min(mx + 2, p.mb_w));
				const uint32_t miw = (uint32_t)__builtin_amdgcn_readfirstlane((int)R.info[mx & (IRW_RING - 1)]);
				const int mb_type = (int)(int8_t)(miw & 0xff);
				const int slot = mx & (IRW_RING - 1);
				const bool left_mb = mx > 0;
				if (mb_type != MB_I4x4) {
					// ---------------- Intra16x16 / I_PCM: the skew has drained; lanes 0..15 = the 16 blocks (coding order) ----------------
					const int l = lane & 15;
					const bool on = lane < 16;
					if (mb_type == MB_IPCM) {
						// its reconstruction is its source (in place already), what its neighbours predict from; the writer knows the rest
						IRW_ORDER();
						if (lane < 4)
							irw_entry_store(entry(gr + 1, 4 * mx + lane), *(const uint32_t *)(&R.io[15][slot * 16 + 4 * lane]));
					} else {
						const int i16_mode = (int)((miw >> 8) & 0xff);
						uint8_t *T = R.t16;
						uint32_t topw = 0, tlw = 0;
						if (top_mb) { // (entries are written left to right: the last one says all)
							wait_entry(gr, 4 * mx + 3);
							topw = (uint32_t)irw_entry_load(entry(gr, 4 * mx + (lane & 3)));
							tlw = (uint32_t)irw_entry_load(entry(gr, 4 * mx - 1));
						}
						if (on) {
							if (left_mb)
								T[(1 + l) * IRW_T16 + 7] = R.io[l][((mx - 1) & (IRW_RING - 1)) * 16 + 15];
							if (top_mb && l < 4)
								*(uint32_t *)(T + 8 + 4 * l) = topw;
							if (top_mb && left_mb && l == 4)
								T[7] = (uint8_t)(tlw >> 24);
						}
						WAVE_SYNC();
						const int bx = blk_x(l), by = blk_y(l);
						int d[16];
						int dcv = 128, pa = 0, pb = 0, pc = 0;
						auto pred16 = [&](int x, int y) {
							return i16_mode == 0 ? (int)T[8 + x]
							     : i16_mode == 1 ? (int)T[(1 + y) * IRW_T16 + 7]
							     : i16_mode == 2 ? dcv
							     : clip255((pa + pb * (x - 7) + pc * (y - 7) + 16) >> 5);
						};
						int16_t *lv = &R.lev[by][(4 * mx + bx) & (IRW_XB - 1)][0];
						if (on) {
							int dcsum = 0, H = 0, V = 0;
							for (int k = 0; k < 16; k++)
								dcsum += (top_mb ? T[8 + k] : 0) + (left_mb ? T[(1 + k) * IRW_T16 + 7] : 0);
							dcv = (top_mb && left_mb) ? (dcsum + 16) >> 5 : (top_mb || left_mb) ? (dcsum + 8) >> 4 : 128;
							for (int k = 0; k < 8; k++) {
								H += (k + 1) * ((int)T[8 + 8 + k] - (int)T[8 + 6 - k]);
								V += (k + 1) * ((int)T[(1 + 8 + k) * IRW_T16 + 7] - (int)T[(1 + 6 - k) * IRW_T16 + 7]);
							}
							pa = 16 * ((int)T[16 * IRW_T16 + 7] + (int)T[8 + 15]);
							pb = (5 * H + 32) >> 6;
							pc = (5 * V + 32) >> 6;
							int w[16];
							{
								int res[16];
#pragma unroll
								for (int i = 0; i < 16; i++) {
									const int x = 4 * bx + (i & 3), y = 4 * by + (i >> 2);
									res[i] = (int)R.io[y][slot * 16 + x] - pred16(x, y);
								}
								fdct4x4(res, w);
							}
							R.dc[bx + 4 * by] = w[0];
							lv[0] = 0;
#pragma unroll
							for (int i = 1; i < 16; i++) {
								const int r = kZigzag[i], c = kPosClass[r];
								const int lq = quant1(w[r], p.mf[c], f_intra, qbits);
								lv[i] = (int16_t)lq;
								d[r] = (lq * p.dv[c]) << p.qshift;
							}
						}
						WAVE_SYNC();
						if (on) { // forward Hadamard output (i, j) = raster lane
							const int i = l >> 2, j = l & 3;
							int s = 0;
#pragma unroll
							for (int a = 0; a < 4; a++)
#pragma unroll
								for (int b = 0; b < 4; b++)
									s += irw_hsign(i, a) * irw_hsign(j, b) * R.dc[4 * a + b];
							R.dcl[l] = quant1((s + 1) >> 1, p.mf[0], 2 * f_intra, qbits + 1);
						}
						WAVE_SYNC();
						if (on) {
							R.dc16[slot][l] = (int16_t)R.dcl[kZigzag[l]];
							int s = 0;
#pragma unroll
							for (int a = 0; a < 4; a++)
#pragma unroll
								for (int b = 0; b < 4; b++)
									s += irw_hsign(by, a) * irw_hsign(bx, b) * R.dcl[4 * a + b];
							const int ls = 16 * p.dv[0];
							d[0] = p.qshift >= 6 ? (s * ls) << (p.qshift - 6) : (s * ls + (1 << (5 - p.qshift))) >> (6 - p.qshift);
							int pred[16], out[16];
#pragma unroll
							for (int i = 0; i < 16; i++)
								pred[i] = pred16(4 * bx + (i & 3), 4 * by + (i >> 2));
							idct4x4_add(d, pred, out);
#pragma unroll
							for (int k = 0; k < 4; k++) // (the tile's inside is only written: predictions read its borders)
								*(uint32_t *)(T + (1 + 4 * by + k) * IRW_T16 + 8 + 4 * bx) =
									(uint32_t)out[4 * k] | (uint32_t)out[4 * k + 1] << 8 | (uint32_t)out[4 * k + 2] << 16 | (uint32_t)out[4 * k + 3] << 24;
						}
						WAVE_SYNC();
						if (on)
							*(u32x4 *)(&R.io[l][slot * 16]) = *(const u32x4 *)(T + (1 + l) * IRW_T16 + 8);
						IRW_ORDER();
						if (lane < 4)
							irw_entry_store(entry(gr + 1, 4 * mx + lane), *(const uint32_t *)(T + 16 * IRW_T16 + 8 + 4 * lane));
					}
					WAVE_SYNC();
					// the skew builds up again from this macroblock's last block column
					rec = R.io[4 * g + qi][slot * 16 + 12 + qj];
					y3 = rowpack(rec);
					mx++;
					continue;
				}

				// ---------------- a run of Intra4x4 macroblocks from mx on: group g is at block column X = X0 + s - 2 g ----------------
				const int X0 = 4 * mx;
				int run_end = nbx; // ... until group 0 finds a macroblock of another kind
				int X = X0 - 2 * g;
				uint32_t ctl1;
				u32x4 sel;
				int srcv;
				uint64_t abn = 0;
				{
					const uint32_t ctl0 = ctl_l[X & (IRW_XB - 1)];
					ctl1 = ctl_l[(X + 1) & (IRW_XB - 1)];
					sel = *(const u32x4 *)(tabl + ctl0);
					srcv = io_l[(4 * X + qj) & (IRW_RB - 1)];
					if (top_mb) {
						const uint32_t a1 = wait_entry(gr, X0 + 1);
						const uint32_t a0 = (uint32_t)irw_entry_load(entry(gr, X0));
						const uint32_t am = (uint32_t)irw_entry_load(entry(gr, X0 - 1));
						if (g == 0) {
							P1 = a0;
							P2 = am;
						}
						abn = (uint64_t)a1 | 1ull << 32;
					}
				}
				for (int s = 0;; s++) {
					const int Xg0 = X0 + s;
					if ((Xg0 & 3) == 0 && Xg0 < run_end) { // group 0 enters a macroblock: staged (and the one after it, for what is fetched ahead)?
						const int m = Xg0 >> 2;
						wait_staged(min(m + 2, p.mb_w));
						const int ty = (int)(int8_t)((uint32_t)__builtin_amdgcn_readfirstlane((int)R.info[m & (IRW_RING - 1)]) & 0xff);
						if (ty != MB_I4x4)
							run_end = Xg0;
					}
					if (Xg0 - 6 >= run_end || xw.dead)
						break;
					// what was fetched a step ago is waited for HERE, on every path: behind the stores of the step nothing
					// may be left for the compiler to wait for (an s_waitcnt there would wait for the stores as well)
					__builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0)
					// the row above group 0's block and the dword right of it: from the row buffer
					uint64_t ab = abn;
					if (top_mb && Xg0 < run_end && Xg0 + 1 < nbx && !(uint32_t)__builtin_amdgcn_readfirstlane((int)(ab >> 32)))
						ab = (uint64_t)wait_entry(gr, Xg0 + 1) | 1ull << 32;
					// ... of the other groups: the bottom row of the block the group above has just finished
					const uint32_t A = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)ab, (int)y3, 0x142, 0xE, 0xF, false); // row_bcast:15
					const uint32_t Lc = colpack(rec);
					const bool act = X >= X0 && X < run_end;
					if (act) {
						const uint32_t pa = __builtin_amdgcn_perm(P2, Lc, sel.x);
						const uint32_t pb = __builtin_amdgcn_perm(A, P1, sel.y);
						const int pv = (int)(__builtin_amdgcn_sad_u8(pb, 0u, __builtin_amdgcn_sad_u8(pa, 0u, sel.z)) >> sel.w);
						const int w = irw_fdct(srcv - pv, ci, cj);
						const int level = irw_quant(w, my_mf, f_intra, c2, qbits);
						lev_l[(X & (IRW_XB - 1)) * 16] = (int16_t)level;
						const int r = irw_idct(__mul24(level, dvq) + rnd32, ci, cj);
						rec = min(max(pv + r, 0), 255);
						io_l[(4 * X + qj) & (IRW_RB - 1)] = (uint8_t)rec;
					}
					P2 = P1;
					P1 = A;
					y3 = rowpack(rec);
					IRW_ORDER();
					if (lane == 63 && act)
						irw_entry_store(entry(gr + 1, X), y3);
					// what the next step reads, fetched now
					X++;
					sel = *(const u32x4 *)(tabl + ctl1);
					ctl1 = ctl_l[(X + 1) & (IRW_XB - 1)];
					srcv = io_l[(4 * X + qj) & (IRW_RB - 1)];
					if (top_mb)
						abn = irw_entry_load(entry(gr, min(Xg0 + 2, nbx)));
				}
				mx = run_end >> 2;
			}
		} else {
			// ------------------------------------------------ chroma ------------------------------------------------
			// lane group g = 4x4 block g of a plane (bx = g & 1, by = g >> 1), Cb then Cr; lane = sample. Prediction reads
			// the row above and the column left of the MACROBLOCK only: no chain inside it.
			IrwRowC &R = S.g[gr];
			const int pclass = LN.pclass, zz = LN.zz;
			const int my_mfc = pclass == 0 ? p.mfc[0] : pclass == 1 ? p.mfc[1] : p.mfc[2];
			const int my_dvc = pclass == 0 ? p.dvc[0] : pclass == 1 ? p.dvc[1] : p.dvc[2];
			const int dvq = my_dvc << p.qshift_c;
			const int qbits = p.qbits_c, f_c = (1 << qbits) / 3, c2 = (1 << qbits) - 1 - 2 * f_c;
			const int bx = g & 1, by = g >> 1;
			const int x = 4 * bx + qj, y = 4 * by + qi;
			uint8_t *T = R.t;
			for (int mx = 0; mx < p.mb_w && !xw.dead; mx++) {
				const int slot = mx & (IRW_RING - 1);
				const bool left_mb = mx > 0;
				wait_staged(mx + 1);
				const uint32_t miw = (uint32_t)__builtin_amdgcn_readfirstlane((int)R.info[slot]);
				const int chroma_mode = (int)((miw >> 16) & 0xff);
				const bool ispcm = (int)(int8_t)(miw & 0xff) == MB_IPCM;
				uint32_t topw = 0, tlw = 0;
				if (top_mb) {
					wait_entry(gr, 4 * mx + 3);
					topw = (uint32_t)irw_entry_load(entry(gr, 4 * mx + (lane & 3)));
					tlw = (uint32_t)irw_entry_load(entry(gr, 4 * mx - 1));
				}
				if (left_mb && lane < 8) // left neighbours = last CbCr pair of the previous macroblock
					*(uint16_t *)(T + (1 + lane) * IRW_TC + 6) = *(const uint16_t *)(T + (1 + lane) * IRW_TC + 22);
				if (top_mb && lane >= 8 && lane < 12)
					*(uint32_t *)(T + 8 + 4 * (lane - 8)) = topw;
				if (top_mb && left_mb && lane == 12)
					*(uint16_t *)(T + 6) = (uint16_t)(tlw >> 16);
				WAVE_SYNC();
				int pvs[2], cdq[2];
				uint32_t ac_any = 0;
#pragma unroll
				for (int c = 0; c < 2; c++) {
					// DC of this block (clause 8.3.4.1-3): 4 samples above / left of the block
					const uint32_t m = 0x00ff00ffu << (8 * c);
					const uint32_t t0 = *(const uint32_t *)(T + 8 + 8 * bx), t1 = *(const uint32_t *)(T + 12 + 8 * bx);
					const int stp = (int)__builtin_amdgcn_sad_u8(t0 & m, 0u, 0u) + (int)__builtin_amdgcn_sad_u8(t1 & m, 0u, 0u);
					const int slf = (int)T[(1 + 4 * by) * IRW_TC + 6 + c] + (int)T[(2 + 4 * by) * IRW_TC + 6 + c] +
							(int)T[(3 + 4 * by) * IRW_TC + 6 + c] + (int)T[(4 + 4 * by) * IRW_TC + 6 + c];
					bool use_t = top_mb, use_l = left_mb;
					if (bx == 1 && by == 0 && top_mb)
						use_l = false;
					if (bx == 0 && by == 1 && left_mb)
						use_t = false;
					const int dcv = (use_t && use_l) ? (stp + slf + 4) >> 3 : use_t ? (stp + 2) >> 2 : use_l ? (slf + 2) >> 2 : 128;
					int pv = dcv;
					if (chroma_mode == 1) {
						pv = T[(1 + y) * IRW_TC + 6 + c];
					} else if (chroma_mode == 2) {
						pv = T[8 + 2 * x + c];
					} else if (chroma_mode == 3) {
						int H = 0, V = 0;
#pragma unroll
						for (int k = 0; k < 4; k++) {
							H += (k + 1) * ((int)T[8 + 2 * (4 + k) + c] - (int)T[8 + 2 * (2 - k) + c]);
							V += (k + 1) * ((int)T[(1 + 4 + k) * IRW_TC + 6 + c] - (int)T[(1 + 2 - k) * IRW_TC + 6 + c]);
						}
						const int pla = 16 * ((int)T[(1 + 7) * IRW_TC + 6 + c] + (int)T[8 + 2 * 7 + c]);
						const int plb = (34 * H + 32) >> 6, plc = (34 * V + 32) >> 6;
						pv = clip255((pla + plb * (x - 3) + plc * (y - 3) + 16) >> 5);
					}
					const int sv = R.io[y][slot * 16 + 2 * x + c];
					const int w = irw_fdct(sv - pv, ci, cj);
					if (LN.dc_lane)
						R.dcw[4 * c + g] = w;
					const int level = (LN.dc_lane || ispcm) ? 0 : irw_quant(w, my_mfc, f_c, c2, qbits);
					R.lev[slot][4 * c + g][zz] = (int16_t)level;
					cdq[c] = __mul24(level, dvq);
					pvs[c] = pv;
					const uint64_t bal = __ballot(level != 0);
					const int nn = __popc((uint32_t)(bal >> (16 * g)) & 0xffffu);
					if (ql == 0)
						R.nnz[slot][4 * c + g] = (uint8_t)(ispcm ? 16 : nn);
					ac_any |= (uint32_t)(bal != 0);
				}
				WAVE_SYNC();
				int cdcnz = 0;
				if (lane < 8) { // 2x2 Hadamard of the chroma DCs: lane = (plane, output index)
					const int c = lane >> 2, blk = lane & 3;
					const int *dcs = R.dcw + 4 * c;
					const int s1 = (blk & 1) ? -1 : 1, s2 = (blk & 2) ? -1 : 1;
					const int hd = dcs[0] + s1 * dcs[1] + s2 * dcs[2] + s1 * s2 * dcs[3];
					const int lq = ispcm ? 0 : quant1(hd, p.mfc[0], 2 * f_c, qbits + 1);
					R.dcl[lane] = lq;
					R.cdc[slot][lane] = (int16_t)lq;
					cdcnz = lq != 0;
				}
				const uint64_t dcm = __ballot(cdcnz != 0);
				WAVE_SYNC();
#pragma unroll
				for (int c = 0; c < 2; c++) {
					const int *ls = R.dcl + 4 * c;
					const int s1 = (g & 1) ? -1 : 1, s2 = (g & 2) ? -1 : 1;
					const int fd = ls[0] + s1 * ls[1] + s2 * ls[2] + s1 * s2 * ls[3];
					const int dcr = ((fd * p.dvc[0] * 16) << p.qshift_c) >> 5;
					const int r = irw_idct((LN.dc_lane ? dcr : cdq[c]) + (LN.dc_lane ? 32 : 0), ci, cj);
					const uint8_t v = ispcm ? R.io[y][slot * 16 + 2 * x + c] : (uint8_t)clip255(pvs[c] + r); // (an I_PCM macroblock: its source sample)
					T[(1 + y) * IRW_TC + 8 + 2 * x + c] = v;
					R.io[y][slot * 16 + 2 * x + c] = v;
				}
				if (lane == 0)
					R.acdc[slot] = ac_any ? 0x20u : (dcm ? 0x10u : 0u);
				WAVE_SYNC();
				IRW_ORDER();
				if (lane < 4)
					irw_entry_store(entry(gr + 1, 4 * mx + lane), *(const uint32_t *)(T + 8 * IRW_TC + 8 + 4 * lane));
				WAVE_SYNC();
			}
		}
		return;
	}

	if (wave == 6) {
		// ======================================== courier ========================================
		// pulls the last row of the band above into row buffer 0: a granule is there when it carries this frame's tag
		int pulled = from_above ? 0 : nbx;
		uint32_t spins = 0;
		uint64_t t0 = 0;
		while (pulled < nbx) {
			const int X = pulled + lane;
			uint64_t v = 0;
			if (X < nbx)
				v = xload64(hb_in + X);
			const uint64_t ok = __ballot(X < nbx && (uint32_t)(v >> 32) == p.db_tag);
			const int n_in = ~ok ? __builtin_ctzll(~ok) : 64; // granules are written left to right: take the prefix that is there
			if (n_in > 0) {
				if (lane < n_in)
					irw_entry_store(entry(0, X), (uint32_t)v);
				pulled += n_in;
				spins = 0;
				t0 = 0;
				continue;
			}
			if (wait_expired(++spins, t0) || ((spins & 15u) == 0 && (xload32(sync + 1) || irw_ctr(&S.abort)))) {
				xstore32(sync + 1, 1u);
				__hip_atomic_store(&S.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_store(p.wf_err, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				break;
			}
			__builtin_amdgcn_s_sleep(4);
		}
		return;
	}

	// ======================================== stager and writer ========================================
	// their 4 lane groups are the 4 rows of the band, each at a macroblock of its own: a row moves on when ITS ring has room
	// (stager) or ITS chain wave has finished a macroblock (writer)
	const int g = lane >> 4, l = lane & 15;
	const int my = y0b + g;
	const bool row_ok = g < nrows;
	const size_t row_mb = (size_t)st * p.n_mb + (size_t)(row_ok ? my : r0) * p.mb_w;
	const bool top_mb = my > r0;

	if (wave == 4) {
		// ======================================== stager ========================================
		const uint8_t *src_y = src_luma(p, st);
		const uint8_t *src_uv = src_chroma(p, st);
		int mx = row_ok ? 0 : p.mb_w; // next macroblock of this lane group's row to stage
		struct StageRegs {
			u32x4 s;
			uint32_t info;
			uint32_t mode;
		};
		auto load = [&](int m, StageRegs &Rg) {
			if (m >= p.mb_w)
				return;
			const size_t a = row_mb + m;
			if (LUMA) {
				Rg.s = *(const u32x4 *)(src_y + (size_t)min(my * 16 + l, p.src_last_y) * p.cw + m * 16);
				Rg.mode = (uint32_t)(uint8_t)p.i4modes[a * 16 + irw_blk_idx(l & 3, l >> 2)];
			} else if (l < 8) {
				Rg.s = *(const u32x4 *)(src_uv + (size_t)min(my * 8 + l, p.src_last_c) * p.cw + m * 16);
			}
			Rg.info = ((const uint32_t *)&p.mbi[a])[0];
		};
		StageRegs Ra{}, Rb{};
		load(mx, Ra);
		load(mx + 1, Rb);
		uint32_t spins = 0;
		uint64_t t0 = 0;
		while (__any(mx < p.mb_w)) {
			// A slot is free once its macroblock is written back -- AND the chain no longer reads it: an Intra16x16 macroblock
			// takes its left column from the ring slot of the macroblock before it (written back the moment the chain is
			// through with it), and the skew restarts from the macroblock's own slot after its row has been handed on. So the
			// ring is used 7 deep: staging macroblock s overwrites s - 8, the chain on macroblock c reads c - 1 and c, the writer
			// has at most c behind it, s <= written + 6 <= c + 6.
			const bool can = mx < p.mb_w && irw_ctr(&S.written[g]) >= mx - IRW_RING + 2;
			if (!__any(can)) {
				if (wait_expired(++spins, t0) || ((spins & 15u) == 0 && irw_ctr(&S.abort))) {
					if (!irw_ctr(&S.abort))
						__hip_atomic_store(p.wf_err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					__hip_atomic_store(&S.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
					xstore32(sync + 1, 1u);
					break;
				}
				__builtin_amdgcn_s_sleep(4);
				continue;
			}
			spins = 0;
			t0 = 0;
			IRW_ORDER();
			if (can) {
				const int slot = mx & (IRW_RING - 1);
				if constexpr (LUMA) {
					IrwRowY &R = S.g[g];
					*(u32x4 *)(&R.io[l][slot * 16]) = Ra.s;
					// the predictor of block (bx, by) = lane: its mode with / without the samples above-right, or DC by
					// what is there (clause 6.4.11.4: availability follows the coding order)
					const int bx = l & 3, by = l >> 2, idx = irw_blk_idx(bx, by);
					const bool left_mb = mx > 0, tr_mb = top_mb && mx < p.mb_w - 1;
					bool ht, hl, htr;
					blk4_avail(idx, top_mb, left_mb, tr_mb, ht, hl, htr);
					const int mode = (int)Ra.mode < 9 ? (int)Ra.mode : 2;
					const int row = mode == 2 ? 18 + (ht ? 1 : 0) + (hl ? 2 : 0) : mode + (htr ? 0 : 9);
					R.ctl[by][(4 * mx + bx) & (IRW_XB - 1)] = (uint16_t)(row * 16 * (int)sizeof(IrwTabEnt));
					if (l == 0)
						R.info[slot] = Ra.info;
				} else {
					IrwRowC &R = S.g[g];
					if (l < 8)
						*(u32x4 *)(&R.io[l][slot * 16]) = Ra.s;
					if (l == 0)
						R.info[slot] = Ra.info;
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
			__builtin_amdgcn_wave_barrier();
			if (can) {
				const bool lost = p.fault_inject == 2 && unit == 0 && band == 0 && mx >= 8 && LUMA; // fault injection (tests): this stager stops reporting
				if (l == 0 && !lost)
					__hip_atomic_store(&S.staged[g], mx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				mx++;
				Ra = Rb;
				load(mx + 1, Rb);
			}
		}
		return;
	}

	// ======================================== writer ========================================
	uint8_t *rec_y = p.rec_y + (size_t)st * p.plane_y_stride;
	uint8_t *rec_uv = p.rec_uv + (size_t)st * p.plane_uv_stride;
	uint32_t *const my_prog = (LUMA ? p.prog_intra : p.prog_intra_c) ? (LUMA ? p.prog_intra : p.prog_intra_c) + unit * PROG_BANDS + (band < PROG_BANDS ? band : PROG_BANDS - 1) : nullptr;
	int mx = row_ok ? 0 : p.mb_w; // next macroblock of this lane group's row to write back
	uint32_t published = 0;
	uint32_t spins = 0;
	uint64_t t0 = 0;
	const bool pushes = feeds_next && g == nrows - 1;
	while (__any(mx < p.mb_w)) {
		const bool can = mx < p.mb_w && (uint32_t)(irw_entry_load(entry(g + 1, 4 * (mx < p.mb_w ? mx : 0) + 3)) >> 32) != 0;
		if (!__any(can)) {
			if (wait_expired(++spins, t0) || ((spins & 15u) == 0 && irw_ctr(&S.abort))) {
				if (!irw_ctr(&S.abort))
					__hip_atomic_store(p.wf_err, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				__hip_atomic_store(&S.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				xstore32(sync + 1, 1u);
				break;
			}
			__builtin_amdgcn_s_sleep(4);
			continue;
		}
		spins = 0;
		t0 = 0;
		IRW_ORDER();
		if (can) {
			const int slot = mx & (IRW_RING - 1);
			const size_t a = row_mb + mx;
			if constexpr (LUMA) {
				IrwRowY &R = S.g[g];
				const uint32_t miw = R.info[slot];
				const int mb_type = (int)(int8_t)(miw & 0xff);
				*(u32x4 *)(rec_y + (size_t)(my * 16 + l) * p.cw + mx * 16) = *(const u32x4 *)(&R.io[l][slot * 16]);
				// lane = block (bx, by): its 16 levels, its count, the Intra16x16 DC level l
				const int bx = l & 3, by = l >> 2, idx = irw_blk_idx(bx, by);
				u32x4 la = u32x4{ 0, 0, 0, 0 }, lb = la;
				if (mb_type != MB_IPCM) {
					const int16_t *lv = &R.lev[by][(4 * mx + bx) & (IRW_XB - 1)][0];
					la = *(const u32x4 *)lv;
					lb = *(const u32x4 *)(lv + 8);
				}
				int16_t *cf = p.coef + a * COEF_PER_MB;
				*(u32x4 *)(cf + COEF_LUMA + idx * 16) = la;
				*(u32x4 *)(cf + COEF_LUMA + idx * 16 + 8) = lb;
				cf[COEF_DC16 + l] = mb_type == MB_I16x16 ? R.dc16[slot][l] : (int16_t)0;
				int nn = 0;
#pragma unroll
				for (int k = 0; k < 4; k++) {
					const uint32_t wa = k == 0 ? la.x : k == 1 ? la.y : k == 2 ? la.z : la.w;
					const uint32_t wb = k == 0 ? lb.x : k == 1 ? lb.y : k == 2 ? lb.z : lb.w;
					nn += ((wa & 0xffffu) != 0) + ((wa >> 16) != 0) + ((wb & 0xffffu) != 0) + ((wb >> 16) != 0);
				}
				if (mb_type == MB_IPCM)
					nn = 16; // (9.2.1: every block of an I_PCM macroblock counts 16 coefficients)
				p.nnz[a * NNZ_PER_MB + bx + 4 * by] = (uint8_t)nn;
				// coded block pattern: an 8x8 quadrant with a level (Intra4x4); all four or none (Intra16x16, AC levels)
				const uint32_t nzb = (uint32_t)(__ballot(nn != 0 && mb_type != MB_IPCM) >> (16 * g)) & 0xffffu; // bit by * 4 + bx
				uint32_t cbp = ((nzb & 0x0033u) ? 1u : 0u) | ((nzb & 0x00ccu) ? 2u : 0u) | ((nzb & 0x3300u) ? 4u : 0u) | ((nzb & 0xcc00u) ? 8u : 0u);
				if (mb_type == MB_I16x16)
					cbp = nzb ? 15u : 0u;
				if (l == 0 && cbp)
					atomicOr((uint32_t *)&p.mbi[a], cbp << 24); // (the chroma workgroups add their bits the same way)
				if (pushes && l < 4)
					xstore64(hb_out + 4 * mx + l, (uint64_t)(*(const uint32_t *)(&R.io[15][slot * 16 + 4 * l])) | (uint64_t)p.db_tag << 32);
			} else {
				IrwRowC &R = S.g[g];
				if (l < 8)
					*(u32x4 *)(rec_uv + (size_t)(my * 8 + l) * p.cw + mx * 16) = *(const u32x4 *)(&R.io[l][slot * 16]);
				int16_t *cf = p.coef + a * COEF_PER_MB;
				*(u32x4 *)(cf + COEF_CAC + (l >> 1) * 16 + (l & 1) * 8) = *(const u32x4 *)(&R.lev[slot][l >> 1][(l & 1) * 8]);
				if (l < 8) {
					cf[COEF_CDC + l] = R.cdc[slot][l];
					p.nnz[a * NNZ_PER_MB + 16 + l] = R.nnz[slot][l];
				}
				const uint32_t cb = R.acdc[slot];
				if (l == 0 && cb)
					atomicOr((uint32_t *)&p.mbi[a], cb << 24);
				if (pushes && l < 4)
					xstore64(hb_out + 4 * mx + l, (uint64_t)(*(const uint32_t *)(&R.io[7][slot * 16 + 4 * l])) | (uint64_t)p.db_tag << 32);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); // (the ring slot has been read)
		__builtin_amdgcn_wave_barrier();
		if (can) {
			if (l == 0)
				__hip_atomic_store(&S.written[g], mx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			mx++;
		}
		// frame pipelining: this picture's deblocking (another kernel, already resident) follows the write-back of every
		// band along the diagonals x + 2*y: every macroblock of the band with x + 2*y < D is in memory
		if (my_prog) {
			int dmin = 0x7fffffff;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const int wk = __builtin_amdgcn_readlane(mx, 16 * k);
				if (k < nrows && wk < p.mb_w)
					dmin = min(dmin, wk + 2 * (y0b + k));
			}
			const uint32_t d = dmin == 0x7fffffff ? (uint32_t)PROG_ALL : (uint32_t)dmin;
			if (d == (uint32_t)PROG_ALL && published != (uint32_t)PROG_ALL && lane == 0)
				dbg_event(p, 4, p.db_tag, (uint32_t)band | (uint32_t)PL << 8 | (uint32_t)unit << 16);
			if (d >= published + 8 || (d == (uint32_t)PROG_ALL && published != (uint32_t)PROG_ALL)) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if (lane == 0)
					__hip_atomic_store(my_prog, p.prog_tag | d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				published = d;
			}
		}
	}
}

// grid: the luma bands of every unit, then the chroma bands (ir_bands workgroups per unit and plane)
__global__ __launch_bounds__(64 * IRW_WAVES) void k_intra_rows(Params p)
{
	const int nwg = p.n_streams * p.slice_num * p.ir_bands;
	if ((int)blockIdx.x < nwg)
		intra_rows_body<1>(p);
	else
		intra_rows_body<2>(p);
}

int launch_intra_rows(const Params &p, void *stream)
{
	hipStream_t s = (hipStream_t)stream;
	const size_t shared = sizeof(IrwShared<1>) > sizeof(IrwShared<2>) ? sizeof(IrwShared<1>) : sizeof(IrwShared<2>);
	const size_t need = shared + (size_t)5 * (4 * p.mb_w + 2) * sizeof(uint64_t);
	constexpr size_t lds_alone = 144 * 1024, lds_beside = 79 * 1024;
	// What a band needs and no more: a pipelined deblock band claims 81 KB so that no second one shares its CU, and this
	// kernel -- which such a band may be waiting for -- must always find room beside it (enc_plan.h). Two of its own
	// workgroups never share a CU: 7 waves of 152 registers. (ir_lds_kb: a larger claim, development.)
	if (need > lds_beside)
		return -1;
	const size_t want = (size_t)p.ir_lds_kb * 1024, lds = want > lds_alone ? lds_alone : want < need ? need : want;
	static DeviceOnce attr;
	if (!attr.run([&]() { return hipFuncSetAttribute((const void *)k_intra_rows, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_alone) == hipSuccess; }))
		return -1;
	const dim3 grid((unsigned)(2 * p.n_streams * p.slice_num * p.ir_bands)), block(64 * IRW_WAVES);
	hipLaunchKernelGGL(k_intra_rows, grid, block, lds, s, p);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace cedar
