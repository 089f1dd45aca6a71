// cedarx_h264_encoder_amd/csrc/enc_k_deblock.hip
// k_deblock_rows: the in-loop deblocking filter of clause 8.7 (raster macroblock order semantics) with ONE WAVE PER
// MACROBLOCK ROW. The dependency chain of deblocking is (x, y) <- (x-1, y), (x, y-1), (x+1, y-1): rows follow each other
// one macroblock apart, 120 + 68 steps for a 1080p picture whatever the kernel does, so the picture costs what ONE step
// costs. In k_deblock2_planes (enc_k_wavefront.hip) a step is ~1100 instructions of one wave: a lane filters a whole
// line of a macroblock, its 4 edges one after the other, and 4 rows share the wave. Here the 4 edges of a line are the 4
// lanes of a quad:
//   * bS < 4 changes at most p1, p0, q0, q1 and reads p2 .. q2, so of edge e-1 only q1' reaches edge e (as its p2), and
//     q1' needs nothing of edge e-2: every lane computes its q1' from the samples as staged, passes it up the quad (DPP),
//     then every lane finishes its edge. The 4 edges cost one edge and a half.
//   * bS 4 (macroblock edges of intra macroblocks) changes q2 as well, which is p1 of edge 1: the strong filter runs
//     first on lane 0 (wave-uniform branch, only where the picture has such edges), lane 1 takes its word from lane 0.
//   * vertical edges: lane (line l, edge e) holds the 4 samples after its edge as one word, the 4 before it come from
//     lane e-1 (lane 0: the last columns of the macroblock to the left, final); horizontal edges: lane (column c, edge e),
//     the same code on words gathered with byte loads.
// A workgroup is ONE band of 4 macroblock rows on one CU, seven waves:
//   4 filter waves  one per row, one per SIMD; they only touch LDS
//   stager          stages samples and edge codes 1 step ahead of the filter waves (4 rows in its 4 lane groups, at step
//                   t group g is at macroblock t - g), with the loads of 5 steps ahead in flight
//   writer          writes filtered tiles back to memory as the rows finish them, publishes the band's progress for
//                   the next frame's analysis, and pushes the bottom rows of the band's last row into the hand-off
//                   buffer as soon as they are final: 8-byte write-through granules, 4 bytes of samples under the frame's tag
//   courier         polls those granules of the band above (the payload says itself that it has arrived: no progress
//                   word) and pulls them into an LDS ring the filter wave of row 0 reads
// Rows inside a band hand their bottom rows on through LDS (`bot`), one macroblock behind each other; a band follows the
// band above by that same macroblock plus two trips through L2 (~3 us), not by steps of its own.
// Luma and chroma are independent given the edge codes: a workgroup filters one plane (chroma: 32 lanes of each wave).
// Every wait is a bounded spin. Bytes are those of k_deblock2_planes (tests/test_gpu_parity.py runs both).
//
// The reference delegates deblocking to silicon (kernel/cedar.c:1135-1153 programs REC/REF/MBINFO addresses, :1176
// triggers); the policy is the slice header's (cedar.c:1024-1029: idc 0, offsets 0).
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "enc_device.h"
#include "enc_wf_common.h"

namespace cedar {

constexpr int DBR_WAVES = 7; // 4 filter waves, stager, writer, courier
constexpr int DBR_RING = 8;  // macroblocks of bottom rows a row keeps for the row (or band) below (power of two)

template <int PL>
struct __attribute__((aligned(16))) DbrRowLds {
	uint8_t y[(PL & 1) ? 4 : 1][(PL & 1) ? 20 * DT_STRIDE : 16]; // tiles, ring by step (layout: enc_wf_common.h)
	uint8_t c[(PL & 2) ? 4 : 1][(PL & 2) ? 10 * DC_STRIDE : 16];
	uint8_t bot[DBR_RING][(PL & 1) ? 4 * 16 : 16];  // last 4 luma rows of macroblock mx (slot mx & 7) ...
	uint8_t botc[DBR_RING][(PL & 2) ? 2 * 16 : 16]; // ... last 2 chroma rows
	uint32_t bsv[(PL & 1) ? 4 : 1][16]; // per sample row: edge codes of the 4 vertical luma edges, one byte each
	uint32_t bsh[(PL & 1) ? 4 : 1][16]; // per sample column: the 4 horizontal luma edges
	uint32_t bsc[(PL & 2) ? 4 : 1][16]; // chroma codes: bytes 0,1 vertical edges 0,2 (index = plane, row); bytes 2,3 horizontal 0,2 (index = column)
	__attribute__((aligned(16))) uint32_t ab[4][4]; // MODE 3: alpha | beta << 8 of the left edge, the same << 16 of the top edge; of the inner edges; chroma likewise
};

template <int PL>
struct DbrShared {
	DbrRowLds<PL> g[4];
	uint8_t top[DBR_RING][(PL & 1) ? 4 * 16 : 2 * 16]; // bottom rows of the band above (courier), slot mx & 7
	int staged;   // steps the stager has staged
	int written;  // steps the writer has written back
	int pushed;   // macroblocks of row 3 whose bottom rows the courier has sent on (their ring slots are free)
	int topready; // macroblocks of the band above in `top`
	int vdone[4]; // macroblocks of row g through their vertical edges
	int filt[4];  // macroblocks of row g filtered
	int ticket;
	int abort;        // a wait of this workgroup ran out: stop waiting
	uint64_t qdb[52]; // MODE 3: per QP alpha, beta, tc0 of bS 1..3, QP_C (one byte each), from the session's table
};

extern __shared__ __attribute__((aligned(16))) uint8_t dbr_lds[];

// lane e of a quad takes the value of lane e-1 (lane 0: its own) / of lane e+1 (lane 3: its own)
DEVI uint32_t quad_prev(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x90, 0xf, 0xf, true); }
DEVI uint32_t quad_next(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xF9, 0xf, 0xf, true); }

// One line of 16 luma samples across its 4 edges, lane e of the quad = edge e. A0: the 4 samples before edge 0 (lane 0
// only: the neighbouring macroblock's, final); B: the 4 samples after the lane's edge, in line order from byte 0.
// code: the lane's edge code (0x80 | tc0, 0x40 on top: bS 4; 0: leave the edge alone); al, be: alpha and beta of the
// lane's edge. SM: 0 no bS 4 in this picture, 1 an IDR picture (a coded edge 0 IS bS 4), 2 bS 4 where the code says so.
// Returns B after all 4 edges; Aout (lane 0) = A0 after edge 0.
template <int SM>
DEVI void edge_quad(uint32_t A0, uint32_t &B, uint32_t &Aout, uint32_t code, int e, int al, int be)
{
	const int valid = -(int)((code >> 7) & 1u);
	int strong = 0;
	if (SM) {
		strong = SM == 1 ? (e == 0 ? valid : 0) : -(int)((code >> 6) & 1u);
		if (__any(strong)) {
			int p3 = (int)(A0 & 0xffu), p2 = (int)((A0 >> 8) & 0xffu), p1 = (int)((A0 >> 16) & 0xffu), p0 = (int)(A0 >> 24);
			int q0 = (int)(B & 0xffu), q1 = (int)((B >> 8) & 0xffu), q2 = (int)((B >> 16) & 0xffu), q3 = (int)(B >> 24);
			filt_luma_s2((uint32_t)strong & 0x80u, 0, al, be, p3, p2, p1, p0, q0, q1, q2, q3);
			A0 = (uint32_t)p3 | (uint32_t)p2 << 8 | (uint32_t)p1 << 16 | (uint32_t)p0 << 24;
			B = (uint32_t)q0 | (uint32_t)q1 << 8 | (uint32_t)q2 << 16 | (uint32_t)q3 << 24;
		}
	}
	const uint32_t Bprev = quad_prev(B); // (every lane takes part: not inside the selection)
	const uint32_t A = e ? Bprev : A0;
	const int nv = valid & ~strong;
	const int tc0 = (int)(code & 0x3fu);
	const int p2 = (int)((A >> 8) & 0xffu), p1 = (int)((A >> 16) & 0xffu), p0 = (int)(A >> 24);
	const int q0 = (int)(B & 0xffu), q1 = (int)((B >> 8) & 0xffu), q2 = (int)((B >> 16) & 0xffu);
	const int m = (max(max(sadb(p0, q0, -al), sadb(p1, p0, -be)), sadb(q1, q0, -be)) >> 31) & nv;
	const int sq = sadb(q2, q0, -be) >> 31;
	const int tcq = tc0 & sq & m;
	const int avg = (p0 + q0 + 1) >> 1;
	const int q1n = q1 + med3i((q2 + avg - (q1 << 1)) >> 1, -tcq, tcq);
	// p2 of edge e is q1 of edge e-1 AFTER that edge (which took nothing from the edge before it)
	const int q1prev = (int)quad_prev((uint32_t)q1n);
	const int p2x = e ? q1prev : p2;
	const int sp = sadb(p2x, p0, -be) >> 31;
	const int tc = (tc0 - sp - sq) & m;
	const int tcp = tc0 & sp & m;
	const int delta = med3i((((q0 - p0) << 2) + (p1 - q1) + 4) >> 3, -tc, tc);
	const int p1n = p1 + med3i((p2x + avg - (p1 << 1)) >> 1, -tcp, tcp);
	const int p0n = med3i(p0 + delta, 0, 255), q0n = med3i(q0 - delta, 0, 255);
	const uint32_t pn = (uint32_t)p1n | (uint32_t)p0n << 8;
	const uint32_t nx = quad_next(pn); // samples 2, 3 after this edge are p1, p0 of the next
	const uint32_t hi = e == 3 ? B >> 16 : nx;
	B = (uint32_t)q0n | (uint32_t)q1n << 8 | hi << 16;
	Aout = (A0 & 0xffffu) | pn << 16;
}

// per-lane variant of lds_wait_ge: every lane has a counter of its own to wait for (the mover's 4 lane groups)
DEVI void lds_wait_ge_lanes(const int *word, int need, const WaitCtx &w)
{
	uint64_t t0 = 0;
	for (int spins = 0; __any(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need); spins++) {
		if (spins < 1024) {
			__builtin_amdgcn_s_sleep(1);
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

// lds_wait_ge with a first phase that sleeps a cycle count instead of re-issuing at once: the waves of a band share 4 SIMDs
// (a filter wave that spins takes issue slots from the stager it waits for)
DEVI void lds_wait_ge1(const int *word, int need, const WaitCtx &w)
{
	uint64_t t0 = 0;
	for (int spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need; spins++) {
		if (spins < 1024) {
			__builtin_amdgcn_s_sleep(1);
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

// MODE as in k_deblock2_planes: 0 a P picture of inter macroblocks; 1 an IDR picture; 2 a P picture with intra
// macroblocks in it; 3 any picture of a session with QPs per macroblock; 4, 5: 0, 1 for pictures that may hold I_PCM
template <int MODE, int PL>
DEVI void deblock_rows_body(const Params &p)
{
	constexpr bool PCM = MODE >= 4;
	constexpr int BASE = PCM ? MODE - 4 : MODE;
	constexpr bool IDR = BASE == 1, MIXED = BASE == 2 || BASE == 3, QPM = BASE == 3;
	constexpr bool DO_Y = (PL & 1) != 0, DO_C = (PL & 2) != 0;
	constexpr int SM = IDR ? 1 : MIXED ? 2 : 0;

	DbrShared<PL> &S = *reinterpret_cast<DbrShared<PL> *>(dbr_lds);
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int NB = p.db_bands;
	uint32_t *sync = p.wf_sync + WF_SYNC_WORDS * (PL == 2 ? 2 : 1);
	if (threadIdx.x == 0) {
		S.ticket = (int)atomicAdd(&sync[0], 1u); // bands are taken in the order workgroups start: whoever waits, waits for a resident one
		S.abort = 0;
		S.staged = 0;
		S.written = 0;
		S.pushed = 0;
		S.topready = 0;
	}
	if (QPM && threadIdx.x < 52) {
		const QpRow &r = p.qtab[threadIdx.x];
		S.qdb[threadIdx.x] = (uint64_t)(uint32_t)r.alpha | (uint64_t)(uint32_t)r.beta << 8 | (uint64_t)(uint32_t)r.tc0[0] << 16 |
				     (uint64_t)(uint32_t)r.tc0[1] << 24 | (uint64_t)(uint32_t)r.tc0[2] << 32 | (uint64_t)(uint32_t)r.qpc << 40;
	}
	if (threadIdx.x < 4) {
		S.vdone[threadIdx.x] = 0;
		S.filt[threadIdx.x] = 0;
	}
	__syncthreads();
	// a ticket beyond the launch means the sync words were not zero when the kernel started: refuse
	// to index memory with it (code 32 in the frame's error word) rather than fault
	if (S.ticket < 0 || S.ticket >= p.n_streams * p.slice_num * NB) {
		if (threadIdx.x == 0)
			__hip_atomic_store(p.wf_err, 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		return;
	}
	const int unit = S.ticket / NB, band = S.ticket % NB;
	const int st = unit / p.slice_num, sl = p.slice_first + unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, sl, &r0, &r1);
	const int nbands = (r1 - r0 + 3) >> 2;
	if (band >= nbands) // (slices of a picture differ by one row: the launch has the bands of the tallest)
		return;
	uint32_t *xabort = sync + 1, *xerr = p.wf_err;
	const WaitCtx xw{xerr, &S.abort, xabort};
	if (threadIdx.x == 0)
		dbg_event(p, 5, p.db_tag, (uint32_t)band | (uint32_t)PL << 8 | (uint32_t)unit << 16);
	const bool feeds_next = band + 1 < nbands; // the last 4 rows of the band's row 3 are stored by whoever filters below
	const bool from_above = band > 0;          // row 0 takes the rows above it from the band above

	if (wave < 4) {
		// =========================== filter wave of row g ===========================
		const int g = wave;
		const int my = r0 + 4 * band + g;
		if (my >= r1)
			return;
		__builtin_amdgcn_s_setprio(3); // the chain: ahead of whatever else shares the SIMD
		const bool top_edge = my > r0;
		DbrRowLds<PL> &L = S.g[g];
		const DbrRowLds<PL> &LU = S.g[g > 0 ? g - 1 : 0];
		const int e = lane & 3, q4 = lane >> 2; // luma: quad = line (vertical edges) / column (horizontal edges), lane of the quad = edge
		for (int mx = 0; mx < p.mb_w; mx++) {
			const int t = mx + g;
			const bool left_edge = mx > 0;
			const int slot = t & 3, pslot = (t + 3) & 3;
			lds_wait_ge1(&S.staged, t + 1, xw);
			// thresholds by edge class: the picture's, or this macroblock's (MODE 3)
			int ayL = p.alpha_y, byL = p.beta_y, ayT = p.alpha_y, byT = p.beta_y, ayI = p.alpha_y, byI = p.beta_y;
			int acL = p.alpha_c, bcL = p.beta_c, acT = p.alpha_c, bcT = p.beta_c, acI = p.alpha_c, bcI = p.beta_c;
			if (QPM) {
				const u32x4 abw = *(const u32x4 *)L.ab[slot];
				ayL = abw.x & 0xff; byL = (abw.x >> 8) & 0xff; ayT = (abw.x >> 16) & 0xff; byT = abw.x >> 24;
				ayI = abw.y & 0xff; byI = (abw.y >> 8) & 0xff;
				acL = abw.z & 0xff; bcL = (abw.z >> 8) & 0xff; acT = (abw.z >> 16) & 0xff; bcT = abw.z >> 24;
				acI = abw.w & 0xff; bcI = (abw.w >> 8) & 0xff;
			}
			if (DO_Y) {
				// ---- vertical edges: quad = sample row q4 ----
				uint8_t *row = L.y[slot] + (4 + q4) * DT_STRIDE + 16;
				uint32_t B = *(const uint32_t *)(row + 4 * e);
				const uint32_t A0 = *(const uint32_t *)(L.y[pslot] + (4 + q4) * DT_STRIDE + 28); // (lane 0 of the quad)
				const uint32_t code = (L.bsv[slot][q4] >> (8 * e)) & 0xffu;
				uint32_t Aout = A0;
				if (__any(code & 0x80u)) {
					edge_quad<SM>(A0, B, Aout, code, e, e ? ayI : ayL, e ? byI : byL);
					*(uint32_t *)(row + 4 * e) = B;
				}
				// the last columns of the macroblock to the left are final now: beside this tile for the write-back,
				// and into its bottom rows for the row below
				if (e == 0 && left_edge) {
					*(uint32_t *)(row - 4) = Aout;
					if (q4 >= 12)
						*(uint32_t *)(L.bot[(mx - 1) & (DBR_RING - 1)] + (q4 - 12) * 16 + 12) = Aout;
				}
			}
			if (DO_C && lane < 32) {
				// ---- vertical chroma edges (x = 0, 4): lane = (row r, edge e2, plane c); nothing passes between them ----
				const int r = lane >> 2, e2 = (lane >> 1) & 1, c = lane & 1;
				uint8_t *row = L.c[slot] + (2 + r) * DC_STRIDE + 16;
				const uint32_t W1 = *(const uint32_t *)(row + 8 * e2);
				const uint32_t W0 = e2 ? *(const uint32_t *)(row + 4) : *(const uint32_t *)(L.c[pslot] + (2 + r) * DC_STRIDE + 28);
				const uint32_t code = (L.bsc[slot][c * 8 + r] >> (8 * e2)) & 0xffu;
				const int p1 = (int)((W0 >> (8 * c)) & 0xffu);
				int p0 = (int)((W0 >> (16 + 8 * c)) & 0xffu), q0 = (int)((W1 >> (8 * c)) & 0xffu);
				const int q1 = (int)((W1 >> (16 + 8 * c)) & 0xffu);
				const int na = e2 ? -acI : -acL, nb = e2 ? -bcI : -bcL;
				const uint32_t strong = SM == 1 ? (e2 == 0 ? code : 0u) : SM == 2 ? ((code & 0x40u) ? code : 0u) : 0u;
				if (SM)
					filt_chroma_s2(strong, 0, na, nb, p1, p0, q0, q1);
				filt_chroma_n2(code ^ strong, 0, na, nb, p1, p0, q0, q1);
				if (e2 || left_edge) {
					row[8 * e2 - 2 + c] = (uint8_t)p0;
					row[8 * e2 + c] = (uint8_t)q0;
				}
				if (e2 == 0 && left_edge) {
					row[-4 + c] = (uint8_t)p1;
					if (r >= 6) {
						uint8_t *bp = L.botc[(mx - 1) & (DBR_RING - 1)] + (r - 6) * 16 + 12;
						bp[c] = (uint8_t)p1;
						bp[2 + c] = (uint8_t)p0;
					}
				}
			}
			lds_signal(&S.vdone[g], mx + 1, lane);
			// rows above: the row above is through (mx, y-1) and through the vertical edges of (mx+1, y-1), the last
			// thing to touch the bottom rows of (mx, y-1); row 0 of a band has them from the courier
			if (g > 0) {
				if (mx + 1 < p.mb_w)
					lds_wait_ge1(&S.vdone[g - 1], mx + 2, xw);
				else
					lds_wait_ge1(&S.filt[g - 1], mx + 1, xw);
			} else if (from_above) {
				lds_wait_ge1(&S.topready, mx + 1, xw);
			} else {
				WAVE_SYNC();
			}
			const int rs = mx & (DBR_RING - 1);
			if (DO_Y) {
				// ---- horizontal edges: quad = sample column q4; the 4 rows above come from the bottom rows of the row above ----
				uint8_t *col = L.y[slot] + 16 + q4;
				const uint8_t *tsrc = (g > 0 ? LU.bot[rs] : S.top[rs]) + q4;
				const uint32_t code = (L.bsh[slot][q4] >> (8 * e)) & 0xffu;
				if (__any(code & 0x80u)) {
					uint8_t *bp = col + (4 + 4 * e) * DT_STRIDE;
					uint32_t B = (uint32_t)bp[0] | (uint32_t)bp[DT_STRIDE] << 8 | (uint32_t)bp[2 * DT_STRIDE] << 16 | (uint32_t)bp[3 * DT_STRIDE] << 24;
					const uint32_t A0 = (uint32_t)tsrc[0] | (uint32_t)tsrc[16] << 8 | (uint32_t)tsrc[32] << 16 | (uint32_t)tsrc[48] << 24;
					uint32_t Aout = A0;
					edge_quad<SM>(A0, B, Aout, code, e, e ? ayI : ayT, e ? byI : byT);
					bp[0] = (uint8_t)B;
					bp[DT_STRIDE] = (uint8_t)(B >> 8);
					bp[2 * DT_STRIDE] = (uint8_t)(B >> 16);
					bp[3 * DT_STRIDE] = (uint8_t)(B >> 24);
					if (e == 0 && top_edge) { // into the tile, which is where the write-back takes them
						col[0] = (uint8_t)Aout;
						col[DT_STRIDE] = (uint8_t)(Aout >> 8);
						col[2 * DT_STRIDE] = (uint8_t)(Aout >> 16);
						col[3 * DT_STRIDE] = (uint8_t)(Aout >> 24);
					}
				} else if (top_edge && lane < 4) {
					*(u32x4 *)(L.y[slot] + lane * DT_STRIDE + 16) = *(const u32x4 *)((g > 0 ? LU.bot[rs] : S.top[rs]) + lane * 16);
				}
			}
			if (DO_C && lane < 32) {
				// ---- horizontal chroma edges (y = 0, 4): lane = (byte column b, edge e2) ----
				const int b = lane >> 1, e2 = lane & 1;
				uint8_t *col = L.c[slot] + 16 + b;
				const uint8_t *pa = e2 ? col + 4 * DC_STRIDE : (g > 0 ? LU.botc[rs] : S.top[rs]) + b;
				const int ps = e2 ? DC_STRIDE : 16;
				uint8_t *qa = col + (2 + 4 * e2) * DC_STRIDE;
				const uint32_t code = (L.bsc[slot][b] >> (16 + 8 * e2)) & 0xffu;
				const int p1 = pa[0];
				int p0 = pa[ps], q0 = qa[0];
				const int q1 = qa[DC_STRIDE];
				const int na = e2 ? -acI : -acT, nb = e2 ? -bcI : -bcT;
				const uint32_t strong = SM == 1 ? (e2 == 0 ? code : 0u) : SM == 2 ? ((code & 0x40u) ? code : 0u) : 0u;
				if (SM)
					filt_chroma_s2(strong, 0, na, nb, p1, p0, q0, q1);
				filt_chroma_n2(code ^ strong, 0, na, nb, p1, p0, q0, q1);
				if (e2 || top_edge) {
					qa[0] = (uint8_t)q0;
					qa[-DC_STRIDE] = (uint8_t)p0; // (edge 0: tile row 1, the last row above)
				}
				if (e2 == 0 && top_edge)
					col[0] = (uint8_t)p1;
			}
			WAVE_SYNC();
			// bottom rows for the row below (their last columns are patched by the next macroblock's vertical edges);
			// row 3's go to the band below: its ring slot must have been sent on
			if (g == 3 && feeds_next && mx >= DBR_RING)
				lds_wait_ge1(&S.pushed, mx - DBR_RING + 1, xw);
			if (DO_Y && lane < 4)
				*(u32x4 *)(L.bot[rs] + lane * 16) = *(const u32x4 *)(L.y[slot] + (16 + lane) * DT_STRIDE + 16);
			if (DO_C && lane < 2)
				*(u32x4 *)(L.botc[rs] + lane * 16) = *(const u32x4 *)(L.c[slot] + (8 + lane) * DC_STRIDE + 16);
			lds_signal(&S.filt[g], mx + 1, lane);
		}
		return;
	}

	const int nsteps = p.mb_w + 3;
	const int y0b = r0 + 4 * band;
	// The bottom rows of a band's last row cross to the band below (another workgroup, another CU) through db_hbuf:
	// 8-byte granules, 4 bytes of samples under this frame's tag, written with write-through atomic stores by the
	// writer as soon as they are final and polled by the courier of the band below -- the payload says itself that it
	// has arrived: no progress word, no waiting for stores to complete before one may be raised, two trips through L2
	// per band instead of four. Luma: granule l = row l >> 2, word l & 3; chroma: granules 16 + (row r, word w) = 16 + 4r + w.
	uint64_t *hb_out = p.db_hbuf + ((size_t)unit * NB + band) * p.mb_w * DBR_HB_GRANULES;
	const uint64_t *hb_in = p.db_hbuf + ((size_t)unit * NB + (band > 0 ? band - 1 : 0)) * p.mb_w * DBR_HB_GRANULES;
	constexpr int NGR = DO_Y ? 16 : 8, GR0 = DO_Y ? 0 : 16; // granules of this plane per macroblock, the first of them
	if (wave == 6) {
		// ============================== courier ==============================
		// up to 4 macroblocks a trip, lane group k = macroblock, lane l of the group = granule
		const int k = lane >> 4, l = lane & 15;
		const bool my_gran = l < NGR;
		int pulled = from_above ? 0 : p.mb_w;
		uint32_t spins = 0;
		uint64_t t0 = 0;
		while (pulled < p.mb_w) {
			// room in the ring: row 0 has filtered what was there 8 macroblocks ago
			const int f0 = __hip_atomic_load(&S.filt[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
			const int room = min(min(p.mb_w, f0 + DBR_RING) - pulled, 4);
			int n_in = 0;
			if (room > 0) {
				uint64_t v = 0;
				const bool mine = k < room && my_gran;
				if (mine)
					v = xload64(hb_in + (size_t)(pulled + k) * DBR_HB_GRANULES + GR0 + l);
				const uint64_t ok = __ballot(!mine || (uint32_t)(v >> 32) == p.db_tag); // a macroblock is there when all its granules are
#pragma unroll
				for (int j = 0; j < 4; j++)
					if (n_in == j && j < room && ((ok >> (16 * j)) & 0xffffu) == 0xffffu)
						n_in = j + 1;
				if (k < n_in && my_gran)
					*(uint32_t *)(S.top[(pulled + k) & (DBR_RING - 1)] + 4 * l) = (uint32_t)v; // (row l >> 2, word l & 3: 16 bytes a row)
			}
			if (n_in > 0) {
				pulled += n_in;
				lds_signal(&S.topready, pulled, lane);
				spins = 0;
				t0 = 0;
				continue;
			}
			if (wait_expired(++spins, t0) || ((spins & 15u) == 0 && (xload32(xabort) || __hip_atomic_load(&S.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)))) {
				xstore32(xabort, 1u);
				__hip_atomic_store(&S.abort, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
				__hip_atomic_store(xerr, 4u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				break;
			}
			if (room > 0)
				__builtin_amdgcn_s_sleep(2);
			else
				__builtin_amdgcn_s_sleep(8);
		}
		return;
	}

	// ============================== stager and writer ==============================
	// their 4 lane groups are the 4 rows of the band: at step t group g is at macroblock t - g (see k_deblock2_planes for
	// the frame pipelining and the two ways of counting progress)
	const int g = lane >> 4, l = lane & 15;
	const int my = r0 + 4 * band + g;
	const bool row_ok = my < r1;
	const bool top_edge = my > r0;
	DbrRowLds<PL> &L = S.g[g];
	uint8_t *rec_y = p.rec_y + (size_t)st * p.plane_y_stride;
	uint8_t *rec_uv = p.rec_uv + (size_t)st * p.plane_uv_stride;
	auto rows_at = [&](const int *ctr, int t) { // every row has its macroblock of step t behind the counter
		const int done = t - g + 1;
		lds_wait_ge_lanes(ctr + g, row_ok ? (done < 0 ? 0 : done > p.mb_w ? p.mb_w : done) : 0, xw);
	};

	if (wave == 5) {
		// ============================== writer ==============================
		uint32_t *const my_prog = p.prog_db ? p.prog_db + ((PL == 2 ? p.n_streams * p.slice_num : 0) + unit) * PROG_BANDS + (band < PROG_BANDS ? band : PROG_BANDS - 1) : nullptr;
		auto publish_diag = [&](uint32_t d) { // every macroblock of this band with x + y < d is written back
			if (!my_prog)
				return;
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			if (lane == 0)
				__hip_atomic_store(my_prog, p.prog_tag | d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		};
		const bool keep_bottom = g == 3 && feeds_next;
		const DbrRowLds<PL> &L3 = S.g[3];
		auto push = [&](int mx) { // the bottom rows of macroblock mx of row 3, final, to the band below (lane group 3: a granule a lane)
			if (g == 3 && l < NGR) {
				const uint32_t w = DO_Y ? *(const uint32_t *)(L3.bot[mx & (DBR_RING - 1)] + 4 * l) : *(const uint32_t *)(L3.botc[mx & (DBR_RING - 1)] + 4 * l);
				xstore64(hb_out + (size_t)mx * DBR_HB_GRANULES + GR0 + l, (uint64_t)w | (uint64_t)p.db_tag << 32);
			}
			lds_signal(&S.pushed, mx + 1, lane); // (read: the ring slot is free)
		};
		for (int t = 0; t < nsteps; t++) {
			rows_at(S.filt, t);
			// row 3 is through macroblock t - 3 and through the vertical edges of it: macroblock t - 4 is final
			if (feeds_next && t >= 4)
				push(t - 4);
			if ((t & p.db_pub_mask) == 0 && t > p.db_pub_mask) // steps < t have been written back: x + y < t + y0b
				publish_diag((uint32_t)(t + y0b));
			const int mx = t - g;
			const bool act = row_ok && mx >= 0 && mx < p.mb_w;
			const bool left_edge = mx > 0;
			const int slot = t & 3;
			const int x0 = mx * 16, y0 = my * 16;
			if (act) {
				if (DO_Y && !(keep_bottom && l >= 12)) {
					*(u32x4 *)(rec_y + (size_t)(y0 + l) * p.cw + x0) = *(const u32x4 *)(L.y[slot] + (4 + l) * DT_STRIDE + 16);
					if (left_edge)
						*(uint32_t *)(rec_y + (size_t)(y0 + l) * p.cw + x0 - 4) = *(const uint32_t *)(L.y[slot] + (4 + l) * DT_STRIDE + 12);
				}
				if (DO_C && l < 8 && !(keep_bottom && l >= 6)) {
					*(u32x4 *)(rec_uv + (size_t)(my * 8 + l) * p.cw + x0) = *(const u32x4 *)(L.c[slot] + (2 + l) * DC_STRIDE + 16);
					if (left_edge)
						*(uint32_t *)(rec_uv + (size_t)(my * 8 + l) * p.cw + x0 - 4) = *(const uint32_t *)(L.c[slot] + (2 + l) * DC_STRIDE + 12);
				}
				if (top_edge) {
					if (DO_Y && l < 4)
						*(u32x4 *)(rec_y + (size_t)(y0 - 4 + l) * p.cw + x0) = *(const u32x4 *)(L.y[slot] + l * DT_STRIDE + 16);
					else if (DO_C && l >= 4 && l < 6)
						*(u32x4 *)(rec_uv + (size_t)(my * 8 - 2 + l - 4) * p.cw + x0) = *(const u32x4 *)(L.c[slot] + (l - 4) * DC_STRIDE + 16);
				}
			}
			lds_signal(&S.written, t + 1, lane); // (the tile has been read: its slot may be staged again)
		}
		if (feeds_next)
			push(p.mb_w - 1); // (nothing to its right patches it)
		publish_diag((uint32_t)PROG_ALL);
		if (lane == 0)
			dbg_event(p, 6, p.db_tag, (uint32_t)band | (uint32_t)PL << 8 | (uint32_t)unit << 16);
		return;
	}

	// ============================== stager ==============================
	const uint32_t cod_y = (0x80u | p.tc0_y[0]) << 8 | (0x80u | p.tc0_y[1]) << 16 | (0x80u | p.tc0_y[2]) << 24;
	const uint32_t cod_c = (0x80u | p.tc0_c[0]) << 8 | (0x80u | p.tc0_c[1]) << 16 | (0x80u | p.tc0_c[2]) << 24;
	AnalysisWait aw{p.prog_inter ? 0u : 0x7fffffffu, IDR || (QPM && p.is_idr), y0b, unit, band, nbands, PL}; // (mode 3 serves the IDR pictures of its sessions too)
	const size_t row_mb = (size_t)st * p.n_mb + (size_t)(row_ok ? my : r0) * p.mb_w;
	struct StageRegs {
		u32x4 in, c;
		uint32_t mv, mvt; // motion-vector dword of MbInfo, kept opaque until staged
		uint32_t ty, tyt; // MIXED: its first dword (type in the low byte)
		uint8_t qc, qt;   // MODE 3: QP the macroblock, and the one above it, is quantised with (qp_t)
		uint8_t nz, nzt;
	};
	StageRegs R0{}, R1{}, R2{}, R3{};
	EdgeCarry carry; // of the macroblock staged last (the left neighbour of the next)

	auto load_step = [&](int t, StageRegs &R) { // global loads for step t
		if (t >= nsteps)
			return;
		const int nmx = t - g;
		const bool nact = row_ok && nmx >= 0 && nmx < p.mb_w;
		const int cmx = nact ? nmx : 0;
		const size_t a = row_mb + cmx;
		const int x0 = cmx * 16, y0 = (row_ok ? my : r0) * 16;
		wait_analysis(p, aw, t, xerr);
		if (DO_Y)
			R.in = *(const u32x4 *)(rec_y + (size_t)(y0 + l) * p.cw + x0);
		if (DO_C)
			R.c = *(const u32x4 *)(rec_uv + (size_t)((y0 >> 1) + (l & 7)) * p.cw + x0);
		if (!IDR) {
			R.mv = ((const uint32_t *)&p.mbi[a])[1];
			R.nz = p.nnz[a * NNZ_PER_MB + l];
			const size_t at = (top_edge && row_ok) ? a - p.mb_w : a;
			R.mvt = ((const uint32_t *)&p.mbi[at])[1];
			R.nzt = p.nnz[at * NNZ_PER_MB + l];
			if (MIXED) {
				R.ty = ((const uint32_t *)&p.mbi[a])[0];
				R.tyt = ((const uint32_t *)&p.mbi[at])[0];
			}
			if (QPM) {
				R.qc = p.qp_t[a];
				R.qt = p.qp_t[at];
			}
		}
		if (PCM) { // pictures that may hold I_PCM macroblocks: the types of this macroblock and the one above
			R.ty = ((const uint32_t *)&p.mbi[a])[0];
			R.tyt = ((const uint32_t *)&p.mbi[(top_edge && row_ok) ? a - p.mb_w : a])[0];
		}
	};

	auto stage_step = [&](int t, const StageRegs &R) { // registers -> tile slot t & 3 + edge codes
		const int mx = t - g;
		const bool act = row_ok && mx >= 0 && mx < p.mb_w;
		const bool left_edge = mx > 0;
		const int slot = t & 3;
		uint32_t nz_cur = 0, nz_top = 0;
		int mvc = 0, mvt = 0;
		if (!IDR) {
			const uint64_t bal_c = __ballot(act && R.nz != 0), bal_t = __ballot(act && top_edge && R.nzt != 0);
			nz_cur = (uint32_t)(bal_c >> (16 * g)) & 0xffffu;
			nz_top = (uint32_t)(bal_t >> (16 * g)) & 0xffffu;
			mvc = (int)R.mv;
			mvt = (int)R.mvt;
		}
		uint32_t wv = 0, wh = 0, wc = 0;
		if (act) {
			if (DO_Y)
				*(u32x4 *)(L.y[slot] + (4 + l) * DT_STRIDE + 16) = R.in;
			if (DO_C && l < 8)
				*(u32x4 *)(L.c[slot] + (2 + l) * DC_STRIDE + 16) = R.c;
			edge_codes<IDR, MIXED, QPM, PCM>(EdgeIn{nz_cur, nz_top, mvc, mvt, R.ty, R.tyt, (int)R.qc, (int)R.qt, mx, l, left_edge, top_edge, cod_y, cod_c},
							 carry, S.qdb, L.ab[slot], wv, wh, wc);
		}
		if (DO_Y) {
			L.bsv[slot][l] = wv;
			L.bsh[slot][l] = wh;
		}
		if (DO_C)
			L.bsc[slot][l] = wc;
	};

	auto stager_iter = [&](int t, StageRegs &R) { // R holds step t+1 and is refilled with step t+5
		if (t + 1 >= nsteps)
			return;
		// slot (t + 1) & 3 held step t - 3: written back, and read as "the tile to the left" by the vertical edges of step t - 2
		if (t >= 3)
			lds_wait_ge1(&S.written, t - 2, xw);
		if (t >= 2)
			rows_at(S.vdone, t - 2);
		stage_step(t + 1, R);
		// fault injection (tests): from step 8 of the first band on, this stager stops reporting
		const bool lost = p.fault_inject == 1 && unit == 0 && band == 0 && t >= 8;
		lds_signal(&S.staged, lost ? 1 : t + 2, lane);
		load_step(t + 5, R); // may wait for this frame's analysis: after the filter waves have their tiles
	};
	// prologue: step 0 staged, loads of steps 1..4 in flight
	load_step(0, R0);
	load_step(1, R1);
	load_step(2, R2);
	load_step(3, R3);
	stage_step(0, R0);
	load_step(4, R0);
	lds_signal(&S.staged, 1, lane);
	for (int t = 0; t < nsteps; t += 4) {
		stager_iter(t, R1);
		stager_iter(t + 1, R2);
		stager_iter(t + 2, R3);
		stager_iter(t + 3, R0);
	}
}

// grid: the luma bands of every unit, then the chroma bands (db_bands workgroups per unit and plane)
template <int MODE>
__global__ __launch_bounds__(64 * DBR_WAVES) void k_deblock_rows(Params p)
{
	const int nwg = p.n_streams * p.slice_num * p.db_bands;
	if ((int)blockIdx.x < nwg)
		deblock_rows_body<MODE, 1>(p);
	else
		deblock_rows_body<MODE, 2>(p);
	if (!p.db_self_reset)
		return;
	// Epilogue as in k_deblock2_planes: the last workgroup to get here leaves the sync words of both deblock regions
	// zeroed for the next frame that uses this slot and tells the host that the frame is deblocked.
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t *r1 = p.wf_sync + WF_SYNC_WORDS, *r2 = p.wf_sync + 2 * WF_SYNC_WORDS;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		if (__hip_atomic_fetch_add(&r1[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			const int used = 16 + p.n_streams * p.slice_num * p.db_bands;
			for (int i = 0; i < used; i++) {
				xstore32(&r1[i], 0u);
				xstore32(&r2[i], 0u);
			}
			if (p.done_flag)
				__hip_atomic_store(p.done_flag, p.done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

int launch_deblock_rows(const Params &p, void *stream)
{
	hipStream_t s = (hipStream_t)stream;
	// db_lds_kb (few units: 96): enough unused dynamic LDS that no second band and no intra workgroup shares the CU -- a
	// filter wave that shares its SIMD with another band's or with the intra wavefront's runs at a fraction of its speed
	constexpr size_t lds_min = sizeof(DbrShared<1>) > sizeof(DbrShared<2>) ? sizeof(DbrShared<1>) : sizeof(DbrShared<2>);
	constexpr size_t lds_alone = 144 * 1024;
	static_assert(lds_min <= 64 * 1024, "a workgroup's tiles");
	const size_t want = (size_t)p.db_lds_kb * 1024, lds = want > lds_alone ? lds_alone : want < lds_min ? lds_min : want;
	static DeviceOnce attr;
	auto dyn = [&](const void *k) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_alone) == hipSuccess; };
	if (!attr.run([&]() {
		    return dyn((const void *)k_deblock_rows<0>) && dyn((const void *)k_deblock_rows<1>) && dyn((const void *)k_deblock_rows<2>) &&
			   dyn((const void *)k_deblock_rows<3>) && dyn((const void *)k_deblock_rows<4>) && dyn((const void *)k_deblock_rows<5>);
	    }))
		return -1;
	const dim3 grid((unsigned)(2 * p.n_streams * p.slice_num * p.db_bands)), block(64 * DBR_WAVES);
	if (p.roi_off) // regions of interest: thresholds per macroblock; an IDR picture is "every macroblock intra" there
		hipLaunchKernelGGL((k_deblock_rows<3>), grid, block, lds, s, p);
	else if (p.pcm && p.is_idr) // I_PCM macroblocks may be among them: their edges are left alone
		hipLaunchKernelGGL((k_deblock_rows<5>), grid, block, lds, s, p);
	else if (p.pcm)
		hipLaunchKernelGGL((k_deblock_rows<4>), grid, block, lds, s, p);
	else if (p.is_idr)
		hipLaunchKernelGGL((k_deblock_rows<1>), grid, block, lds, s, p);
	else if (p.intra_refresh > 0) // intra macroblocks among the inter ones: both sets of boundary strengths
		hipLaunchKernelGGL((k_deblock_rows<2>), grid, block, lds, s, p);
	else
		hipLaunchKernelGGL((k_deblock_rows<0>), grid, block, lds, s, p);
	return hipGetLastError() == hipSuccess ? 0 : -1;
}

} // namespace cedar
