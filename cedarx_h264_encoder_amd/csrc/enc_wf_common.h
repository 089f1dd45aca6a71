// cedarx_h264_encoder_amd/csrc/enc_wf_common.h
// What the wavefront kernels share (enc_k_wavefront.hip: intra reconstruction and the band-pair deblocking; enc_k_deblock.hip:
// deblocking with one wave per macroblock row): LDS-only wave synchronisation, the write-through hand-off between
// workgroups, bounded LDS waits, the deblocking filters of clause 8.7 on one line of samples, the per-row LDS tiles.
#pragma once
#include <mutex>

#include <hip/hip_runtime.h>

#include "enc_device.h"

namespace cedar {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// Host side of the launchers: hipFuncSetAttribute applies to the CURRENT device's copy of a kernel, so "done" is kept per
// device, and sessions may be configured from several threads, so it is kept under a lock.
struct DeviceOnce {
	std::mutex m;
	bool done[64] = {};
	template <typename F> bool run(F set)
	{
		int dev = 0;
		if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64)
			return false;
		std::lock_guard<std::mutex> g(m);
		if (!done[dev]) {
			if (!set())
				return false;
			done[dev] = true;
		}
		return true;
	}
};

// intra-wave hand-off through LDS only: order LDS accesses, leave global loads
// and stores in flight (a plain fence would drain them with vmcnt(0))
#define WAVE_SYNC()                                                            \
	do {                                                                   \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
		__builtin_amdgcn_wave_barrier();                               \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); \
	} while (0)

// ---- hand-off between the workgroups (CUs) that share one unit's wavefront ----
// Producer: payload as 8-byte write-through (agent-scope relaxed atomic = sc1)
// stores, every storing wave drains with s_waitcnt vmcnt(0), then ONE lane stores
// the progress word. Consumer: relaxed agent-scope poll of that ONE word, then
// sc1 loads of the payload (cdna guide G16, form R1, first row of the validity
// table). Spins are bounded: a time-out raises the abort word, which every waiter
// checks, and flags the frame as failed instead of hanging the GPU.
DEVI uint32_t xload32(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVI void xstore32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVI uint64_t xload64(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
DEVI void xstore64(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wave-uniform: wait until *word >= need; returns the value seen (or a huge one after abort)
DEVI uint32_t xwait(const uint32_t *word, uint32_t *abort_word, uint32_t *err_word, uint32_t need)
{
	uint32_t v = xload32(word), spins = 0;
	uint64_t t0 = 0;
	while (v < need) {
		if (wait_expired(++spins, t0) || xload32(abort_word)) {
			xstore32(abort_word, 1u);
			__hip_atomic_store(err_word, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			return 0x7fffffffu;
		}
		__builtin_amdgcn_s_sleep(8);
		v = xload32(word);
	}
	return v;
}

// ------------------------------------------------------------------------
// k_deblock2: in-loop deblocking as a macroblock wavefront. A band of 4 macroblock
// rows is swept left to right; lane group g = lane>>4 works on row band+g, two
// macroblocks behind group g-1, so all 64 lanes are busy. Within a group the 16
// lanes are the 16 sample rows of the macroblock for the vertical edges and its 16
// sample columns for the horizontal edges; each lane filters its line in registers
// with pure integer mask arithmetic (hipcc turns `c ? a : b` chains into exec-mask
// branches, which saturate the CU's one scalar pipe).
//
// SPECIALISED WAVES. A wave alone on its SIMD issues one instruction per ~4.5
// cycles whatever their dependencies, so a step costs its instruction count. One
// wave doing everything needed ~1400 instructions per step, of which ~550 filter
// samples. Here every band is served by a PAIR:
//   filter wave  per step: copy left columns / top rows inside LDS, 4 vertical +
//                4 horizontal edges on the staged tile, save the bottom rows (`bot`)
//   mover wave   ahead: loads of step t+3 in flight, stages step t+1 (samples +
//                edge codes) into the next of 3 tile slots; behind: writes step t-1
//                back to memory, feeds the next workgroup, publishes progress
// They meet only through two monotonic LDS counters (staged / filtered). The bottom
// rows a macroblock hands to the row below travel through LDS (`bot`) between the
// groups of one filter wave; band boundaries (another pair) go through memory under
// the progress counters, workgroup boundaries through the write-through hand-off
// buffer. Templates: picture type (P pictures carry no strong-filter code) and planes
// (luma and chroma are independent given the edge codes and run as two kernels).
// ------------------------------------------------------------------------
#define DT_STRIDE 32 // luma tile: (x,y) at [(4+y)*32 + 16 + x], x,y in [-4,15]
#define DC_STRIDE 32 // chroma tile (nv12): (x,y,c) at [(2+y)*32 + 16 + 2x + c], x,y in [-2,7]
// Edge codes: the mover turns every boundary strength into one byte, 0x80 | tc0 (0 = edge
// not filtered), so the filter wave spends two bit-field extracts on it. |a-b|+c on byte
// values is one v_sad_u8; thresholds go in negated so the sign bit is the comparison.
DEVI int sadb(int a, int b, int c) { return (int)__builtin_amdgcn_sad_u8((uint32_t)a, (uint32_t)b, (uint32_t)c); }
DEVI int med3i(int x, int lo, int hi) { return min(max(x, lo), hi); }
DEVI int code_tc0(uint32_t w, int e) { return (int)((w >> (8 * e)) & 0x7fu); }
DEVI int code_valid(uint32_t w, int e) { return (int)(w << (24 - 8 * e)) >> 31; }

DEVI void filt_luma_n2(uint32_t w, int e, int nalpha, int nbeta,
		       int &p2, int &p1, int &p0, int &q0, int &q1, int &q2)
{
	const int tc0 = code_tc0(w, e);
	const int m = (max(max(sadb(p0, q0, nalpha), sadb(p1, p0, nbeta)), sadb(q1, q0, nbeta)) >> 31) & code_valid(w, e);
	const int sp = sadb(p2, p0, nbeta) >> 31, sq = sadb(q2, q0, nbeta) >> 31; // -1 where the side is smooth
	const int tc = (tc0 - sp - sq) & m;
	const int tcp = tc0 & sp & m, tcq = tc0 & sq & m; // a masked-off update is a clip to [0, 0]
	const int delta = med3i((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc);
	const int avg = (p0 + q0 + 1) >> 1;
	p1 += med3i((p2 + avg - (p1 << 1)) >> 1, -tcp, tcp);
	q1 += med3i((q2 + avg - (q1 << 1)) >> 1, -tcq, tcq);
	p0 = med3i(p0 + delta, 0, 255);
	q0 = med3i(q0 - delta, 0, 255);
}

DEVI void filt_luma_s2(uint32_t w, int e, int alpha, int beta, int p3, int &p2, int &p1, int &p0,
		       int &q0, int &q1, int &q2, int q3)
{
	const int d = sadb(p0, q0, 0);
	const int m = (max(max(d - alpha, sadb(p1, p0, -beta)), sadb(q1, q0, -beta)) >> 31) & code_valid(w, e);
	const int small = (d - ((alpha >> 2) + 2)) >> 31;
	const int sp = small & (sadb(p2, p0, -beta) >> 31), sq = small & (sadb(q2, q0, -beta) >> 31);
	const int np0 = sel(sp, (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, (2 * p1 + p0 + q1 + 2) >> 2);
	const int np1 = sel(sp, (p2 + p1 + p0 + q0 + 2) >> 2, p1);
	const int np2 = sel(sp, (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3, p2);
	const int nq0 = sel(sq, (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, (2 * q1 + q0 + p1 + 2) >> 2);
	const int nq1 = sel(sq, (p0 + q0 + q1 + q2 + 2) >> 2, q1);
	const int nq2 = sel(sq, (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3, q2);
	p0 = sel(m, np0, p0); p1 = sel(m, np1, p1); p2 = sel(m, np2, p2);
	q0 = sel(m, nq0, q0); q1 = sel(m, nq1, q1); q2 = sel(m, nq2, q2);
}

DEVI void filt_chroma_n2(uint32_t w, int e, int nalpha, int nbeta, int p1, int &p0, int &q0, int q1)
{
	const int m = (max(max(sadb(p0, q0, nalpha), sadb(p1, p0, nbeta)), sadb(q1, q0, nbeta)) >> 31) & code_valid(w, e);
	const int tc = (code_tc0(w, e) + 1) & m;
	const int delta = med3i((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc);
	p0 = med3i(p0 + delta, 0, 255);
	q0 = med3i(q0 - delta, 0, 255);
}

DEVI void filt_chroma_s2(uint32_t w, int e, int nalpha, int nbeta, int p1, int &p0, int &q0, int q1)
{
	const int m = (max(max(sadb(p0, q0, nalpha), sadb(p1, p0, nbeta)), sadb(q1, q0, nbeta)) >> 31) & code_valid(w, e);
	const int np0 = (2 * p1 + p0 + q1 + 2) >> 2, nq0 = (2 * q1 + q0 + p1 + 2) >> 2;
	p0 = sel(m, np0, p0);
	q0 = sel(m, nq0, q0);
}

// byte e of an edge-code word if it carries the bS 4 mark (0x40 on top of 0x80), else 0
DEVI uint32_t strong_part(uint32_t w, int e)
{
	return ((w >> (8 * e)) & 0x40u) ? w & (0xffu << (8 * e)) : 0u;
}

// packed { int16 mvx, mvy } words: 1 if a component differs by 4 quarter samples or more
DEVI int mv_far(int a, int b)
{
	const int dx = (int)(int16_t)a - (int)(int16_t)b, dy = (a >> 16) - (b >> 16);
	return (int)((uint32_t)(3 - iabs(dx)) >> 31) | (int)((uint32_t)(3 - iabs(dy)) >> 31);
}

// (PL: the planes a workgroup filters, 1 luma, 2 chroma, 3 both. A one-plane workgroup only carries its plane's
// tiles: 2.6 KB resp. 1.3 KB per row group instead of 3.9 -- in the small form that is what lets it into the
// room ONE retiring k_inter workgroup leaves on a full CU, 22 KB + the CU's 7.7 KB of slack)
template <int PL>
struct __attribute__((aligned(16))) Db2GroupLds {
	uint8_t y[(PL & 1) ? 3 : 1][(PL & 1) ? 20 * DT_STRIDE : 16];
	uint8_t c[(PL & 2) ? 3 : 1][(PL & 2) ? 10 * DC_STRIDE : 16];
	uint8_t bot[4][(PL & 1) ? 4 * 16 : 16];
	uint8_t botc[4][(PL & 2) ? 2 * 16 : 16];
	uint32_t bsv[(PL & 1) ? 3 : 1][16];  // per lane (= sample row): edge codes of the 4 vertical luma edges, one byte each
	uint32_t bsh[(PL & 1) ? 3 : 1][16];  // per lane (= sample column): the 4 horizontal luma edges
	uint32_t bsc[(PL & 2) ? 3 : 1][16];  // chroma codes: bytes 0,1 vertical edges 0,2 (lane = plane, row); bytes 2,3 horizontal 0,2 (lane = column)
	__attribute__((aligned(16))) uint32_t ab[3][4]; // MODE 3: alpha | beta << 8 of the left edge, the same << 16 of the top edge; of the inner edges; chroma likewise
};

// Where a wave of a wavefront workgroup reports a wait that ran out, and where it looks for the
// reports of others: once one hand-off is lost every later wait would run out too, one after the
// other; the abort flags make the rest of the grid drain in milliseconds instead.
struct WaitCtx {
	uint32_t *err;     // pinned host word of the frame (collect turns it into -EIO)
	int *abort_lds;    // this workgroup
	uint32_t *abort_x; // all workgroups of the launch (device memory, zeroed per launch when wf_split > 1)
};

DEVI void lds_wait_ge(const int *word, int need, const WaitCtx &w)
{
	// bounded like every spin here: a lost partner ends in -EIO at collect, not in a hung grid
	uint64_t t0 = 0;
	for (int spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need; spins++) {
		// back off after ~50 us: the partner is usually a few hundred cycles away, but with frames
		// pipelined a whole workgroup can wait for a millisecond for its frame's analysis
		if (spins < 2048) {
			__builtin_amdgcn_s_sleep(0);
			continue;
		}
		if (wait_expired((uint32_t)spins, t0)) {
			__hip_atomic_store(w.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(w.abort_lds, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			xstore32(w.abort_x, 1u);
			break;
		}
		if (__hip_atomic_load(w.abort_lds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP))
			break;
		__builtin_amdgcn_s_sleep(32);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

DEVI void lds_signal(int *word, int value, int lane)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
	__builtin_amdgcn_wave_barrier();
	if (lane == 0)
		__hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- edge codes of one macroblock (the stagers of both deblock kernels) ----
// What a stager carries along its row: the macroblock staged last is the left neighbour of the next.
struct EdgeCarry {
	uint32_t nz_left = 0; // its coded-block bits
	int mvl = 0;          // its motion vector dword
	bool intra_left = false, pcm_left = false;
	int eff_left = 0, teff_left = 0; // MODE 3: QP_Y the decoder holds for it, and for the one above it
};
struct EdgeIn {
	uint32_t nz_cur, nz_top; // coded-block bits of the macroblock (bit 4*by + bx) and of the one above
	int mvc, mvt;            // motion vector dwords of the macroblock and of the one above
	uint32_t ty, tyt;        // first dword of their records (type in the low byte, coded block pattern in the high one)
	int qc, qt;              // MODE 3: QP they are quantised with
	int mx, l;               // macroblock column; lane of the row's group (sample row / column, chroma index)
	bool left_edge, top_edge;
	uint32_t cod_y, cod_c;   // the picture's edge code by boundary strength 1..3 in bytes 1..3 (luma, chroma)
};
// wv / wh: per sample row / column l the edge codes of the 4 vertical / horizontal luma edges, one byte each (0x80 | tc0,
// 0x40 on top: bS 4, 0: leave the edge alone); wc: chroma, bytes 0,1 vertical edges 0,2 (l = plane, row), bytes 2,3
// horizontal 0,2 (l = column). MODE 3 also leaves the macroblock's alpha / beta per edge class in ab_slot (lane 0).
template <bool IDR, bool MIXED, bool QPM, bool PCM>
DEVI void edge_codes(const EdgeIn &in, EdgeCarry &cy, const uint64_t *qdb, uint32_t *ab_slot, uint32_t &wv, uint32_t &wh, uint32_t &wc)
{
	const uint32_t nz_cur = in.nz_cur, nz_top = in.nz_top, cod_y = in.cod_y, cod_c = in.cod_c;
	const int mvc = in.mvc, mvt = in.mvt, mx = in.mx, l = in.l;
	const bool left_edge = in.left_edge, top_edge = in.top_edge;
	const struct { uint32_t ty, tyt; int qc, qt; } R{in.ty, in.tyt, in.qc, in.qt};
	uint32_t &nz_left = cy.nz_left;
	int &mvl = cy.mvl, &eff_left = cy.eff_left, &teff_left = cy.teff_left;
	bool &intra_left = cy.intra_left, &pcm_left = cy.pcm_left;
	if (IDR) { // all intra: bS 4 (strong) on macroblock edges, 3 inside
		const uint32_t in_y = (cod_y >> 24) * 0x01010100u, in_c = cod_c >> 24;
		wv = in_y | (left_edge ? 0x80u : 0u);
		wh = in_y | (top_edge ? 0x80u : 0u);
		wc = (left_edge ? 0x80u : 0u) | in_c << 8 | (top_edge ? 0x80u : 0u) << 16 | in_c << 24;
	} else {
		// edge code by strength for the macroblock's left edge, top edge, inner edges (MODE 3: its own tc0)
		uint32_t cyL = cod_y, cyT = cod_y, cyI = cod_y, ccL = cod_c, ccT = cod_c, ccI = cod_c;
		if (QPM) {
			// QP_Y as the decoder holds it: a macroblock without mb_qp_delta keeps that of the one before it in
			// its row; thresholds from the mean across a macroblock edge (8.7.2.2), per plane
			auto coded = [](uint32_t w0) { return (int)(int8_t)(w0 & 0xffu) == MB_I16x16 || ((w0 >> 24) & 0x3fu) != 0; };
			const int eff = (mx == 0 || coded(R.ty)) ? (int)R.qc : eff_left;
			const int teff = (mx == 0 || coded(R.tyt)) ? (int)R.qt : teff_left;
			auto qpc_of = [&](int q) { return (int)((qdb[q] >> 40) & 0xffu); };
			const int ce = qpc_of(eff);
			const uint64_t yL = qdb[(eff_left + eff + 1) >> 1], yT = qdb[(teff + eff + 1) >> 1], yI = qdb[eff];
			const uint64_t cL = qdb[(qpc_of(eff_left) + ce + 1) >> 1], cT = qdb[(qpc_of(teff) + ce + 1) >> 1], cI = qdb[ce];
			auto code3 = [](uint64_t r) { // tc0 of bS 1..3 -> edge codes 0x80 | tc0 in bytes 1..3
				return (uint32_t)((r >> 8) & 0xffffff00ull) | 0x80808000u;
			};
			cyL = code3(yL); cyT = code3(yT); cyI = code3(yI);
			ccL = code3(cL); ccT = code3(cT); ccI = code3(cI);
			if (l == 0)
				*(u32x4 *)ab_slot = u32x4{ (uint32_t)(yL & 0xffffu) | (uint32_t)(yT & 0xffffu) << 16, (uint32_t)(yI & 0xffffu),
							      (uint32_t)(cL & 0xffffu) | (uint32_t)(cT & 0xffffu) << 16, (uint32_t)(cI & 0xffffu) };
			eff_left = eff;
			teff_left = teff;
		}
		const int k = l >> 2, ks = (l & 7) >> 1;
		int bv[4], bh[4];
		// vectors are in quarter samples: bS 1 from a difference of a whole sample on (8.7.2.1)
		const int mvd_l = mv_far(mvc, mvl), mvd_t = mv_far(mvc, mvt);
		const int nzv0 = (int)(((nz_cur >> (4 * k)) | (nz_left >> (4 * k + 3))) & 1u);
		bv[0] = (2 * nzv0 + ((1 - nzv0) & mvd_l)) & -(int)left_edge;
		const int nzh0 = (int)(((nz_cur >> k) | (nz_top >> (12 + k))) & 1u);
		bh[0] = (2 * nzh0 + ((1 - nzh0) & mvd_t)) & -(int)top_edge;
#pragma unroll
		for (int e = 1; e < 4; e++) {
			bv[e] = 2 * (int)(((nz_cur >> (4 * k + e)) | (nz_cur >> (4 * k + e - 1))) & 1u);
			bh[e] = 2 * (int)(((nz_cur >> (4 * e + k)) | (nz_cur >> (4 * e - 4 + k))) & 1u);
		}
		const int nzc0 = (int)(((nz_cur >> (4 * ks)) | (nz_left >> (4 * ks + 3))) & 1u);
		const int bsc0 = (2 * nzc0 + ((1 - nzc0) & mvd_l)) & -(int)left_edge;
		const int bsc2 = 2 * (int)(((nz_cur >> (4 * ks + 2)) | (nz_cur >> (4 * ks + 1))) & 1u);
#pragma unroll
		for (int e = 0; e < 4; e++) {
			wv |= (((e == 0 ? cyL : cyI) >> (8 * bv[e])) & 0xffu) << (8 * e);
			wh |= (((e == 0 ? cyT : cyI) >> (8 * bh[e])) & 0xffu) << (8 * e);
		}
		wc = ((ccL >> (8 * bsc0)) & 0xffu) | ((ccI >> (8 * bsc2)) & 0xffu) << 8 |
		     ((ccT >> (8 * bh[0])) & 0xffu) << 16 | ((ccI >> (8 * bh[2])) & 0xffu) << 24;
		if (MIXED) { // an intra macroblock on either side: bS 4 on the macroblock edge (code 0xC0), 3 inside
			const bool ic = (int)(int8_t)(R.ty & 0xffu) < MB_P16x16, it = (int)(int8_t)(R.tyt & 0xffu) < MB_P16x16;
			if (ic) {
				const uint32_t in_y = (cyI >> 24) * 0x01010100u, in_c = ccI >> 24;
				wv = (wv & 0xffu) | in_y;
				wh = (wh & 0xffu) | in_y;
				wc = (wc & 0x00ff00ffu) | in_c << 8 | in_c << 24;
			}
			if (left_edge && (ic || intra_left)) {
				wv = (wv & ~0xffu) | 0xC0u;
				wc = (wc & ~0xffu) | 0xC0u;
			}
			if (top_edge && (ic || it)) {
				wh = (wh & ~0xffu) | 0xC0u;
				wc = (wc & ~0xff0000u) | 0xC00000u;
			}
			intra_left = ic;
		}
	}
	if (PCM) {
		// I_PCM (only pictures at whose QP no edge of such a macroblock is filtered at all): its own edges, inner
		// ones included, and its neighbours' edges towards it are left alone
		const bool pc = (int)(int8_t)(R.ty & 0xffu) == MB_IPCM, pt = top_edge && (int)(int8_t)(R.tyt & 0xffu) == MB_IPCM;
		if (pc)
			wv = wh = wc = 0;
		if (pcm_left) {
			wv &= ~0xffu;
			wc &= ~0xffu;
		}
		if (pt) {
			wh &= ~0xffu;
			wc &= ~0xff0000u;
		}
		pcm_left = pc;
	}
	nz_left = nz_cur;
	mvl = mvc;
}

// ---- a deblock band waits for its frame's analysis (the loading wave of both deblock kernels) ----
// Frame pipelining: this frame's analysis runs ahead on another stream in diagonal stripes, or -- an IDR picture -- as the
// intra wavefront, and publishes how far it is. A band of 4 rows sweeps with ONE macroblock of lag per row: at step t its
// rows are at x + y = t + y0b, which is what a P picture's analysis (stripes of x + y, published by k_gate) is asked
// for. The intra wavefront needs TWO macroblocks of lag per row (x + 2*y) and publishes per band in that measure; intra
// prediction reads UNFILTERED neighbours and deblocking works in place, so in an intra picture the deblock wavefront
// also stays behind (x+1, y+1): three more diagonals. `seen` caches what the analysis was last known to have reached.
struct AnalysisWait {
	uint32_t seen;      // diagonals of this frame known to be analysed (0x7fffffff: frames are not pipelined, or a wait ran out)
	bool idr_pic;       // the picture is all intra: ask the intra wavefront's band words, in x + 2*y
	int y0b, unit, band, nbands;
	int planes = 3;     // what the waiting workgroup filters: 1 luma, 2 chroma, 3 both (k_intra_rows publishes the planes apart)
};
DEVI void wait_analysis(const Params &p, AnalysisWait &a, int t, uint32_t *xerr)
{
	const uint32_t need = a.idr_pic ? (uint32_t)(t + 3 + 2 * a.y0b + 4)  // x + 2*y: group 3 is 3 diagonals ahead of group 0
					: (uint32_t)(t + a.y0b + 1);         // x + y: all four rows are on t + y0b
	if (a.seen >= need)
		return;
	// P pictures: one word, the analysis runs in whole diagonal stripes. IDR pictures: the intra wavefront publishes per
	// band; this band, the one above (its rows above) and the one below (its consumers) count.
	const uint32_t *w0 = p.prog_inter, *w1 = p.prog_inter, *w2 = p.prog_inter;
	// (k_intra_rows: luma and chroma are wavefronts of their own; a workgroup that filters both planes asks both)
	const bool two = a.idr_pic && p.prog_intra && p.prog_intra_c && a.planes == 3;
	const ptrdiff_t cdist = two ? p.prog_intra_c - p.prog_intra : 0;
	if (a.idr_pic && p.prog_intra) {
		const uint32_t *base = p.prog_intra_c && a.planes == 2 ? p.prog_intra_c : p.prog_intra;
		w0 = base + a.unit * PROG_BANDS + (a.band < PROG_BANDS ? a.band : PROG_BANDS - 1);
		w1 = a.band > 0 ? w0 - 1 : w0;
		w2 = a.band + 1 < a.nbands && a.band + 1 < PROG_BANDS ? w0 + 1 : w0;
	}
	uint32_t spins = 0;
	uint64_t t0 = 0;
	while (true) {
		uint32_t v0 = __hip_atomic_load(w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		uint32_t v1 = __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		uint32_t v2 = __hip_atomic_load(w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t want = p.prog_tag | need;
		if (two) { // the lesser of the two planes' words (same tag or an older one: compared as distances from `want`)
			const uint32_t c0 = __hip_atomic_load(w0 + cdist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t c1 = __hip_atomic_load(w1 + cdist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			const uint32_t c2 = __hip_atomic_load(w2 + cdist, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			v0 = (int32_t)(c0 - want) < (int32_t)(v0 - want) ? c0 : v0;
			v1 = (int32_t)(c1 - want) < (int32_t)(v1 - want) ? c1 : v1;
			v2 = (int32_t)(c2 - want) < (int32_t)(v2 - want) ? c2 : v2;
		}
		if ((int32_t)(v0 - want) >= 0 && (int32_t)(v1 - want) >= 0 && (int32_t)(v2 - want) >= 0) {
			const uint32_t x = v0 & 0xffffu, y = v1 & 0xffffu, z = v2 & 0xffffu;
			a.seen = x < y ? (x < z ? x : z) : (y < z ? y : z);
			break;
		}
		// (a 3 us sleep per look: the clock is read every 256th, i.e. every ~0.8 ms)
		if (wait_expired(++spins, t0) || ((spins & 63u) == 0 && __hip_atomic_load(p.abort_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
			__hip_atomic_store(xerr, 16u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			// (development, CEDAR_DEBUG_HANG: the first wait that runs out says where it stood)
			if ((threadIdx.x & 63) == 0 && atomicCAS(p.abort_all + 1, 0u, 1u + (uint32_t)a.band + ((uint32_t)a.planes << 8) + ((uint32_t)a.idr_pic << 12) + ((uint32_t)a.unit << 16)) == 0u) {
				p.abort_all[2] = need;
				p.abort_all[3] = __hip_atomic_load(w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				p.abort_all[4] = __hip_atomic_load(w1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				p.abort_all[5] = __hip_atomic_load(w2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				p.abort_all[6] = (uint32_t)t;
				p.abort_all[7] = p.prog_tag;
			}
			__hip_atomic_store(p.abort_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			a.seen = 0x7fffffffu;
			break;
		}
		// a kernel that waits for its frame's analysis is resident for a long time, and every poll is a trip to memory:
		// after a few quick looks, look every ~3 us
		if (spins < 8)
			__builtin_amdgcn_s_sleep(16);
		else
			__builtin_amdgcn_s_sleep(127);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

} // namespace cedar
