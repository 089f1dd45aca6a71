// cedarx_h264_encoder_amd/csrc/enc_k_wavefront.hip
// The two macroblock-ORDERED stages of H.264 as gfx950 wavefront kernels:
//   k_intra_recon2     closed-loop intra prediction + transform + reconstruction
//                      (needs left / top / top-right reconstructed neighbours)
//   k_deblock2_planes  in-loop deblocking filter, clause 8.7, raster MB order
//                      semantics (an MB's filters read samples its left / top /
//                      top-right neighbours have already filtered)
// Both have the dependency (x, y) <- (x-1, y), (x, y-1), (x+1, y-1): an anti-diagonal
// wavefront, a 2-MB lag per row. A band of 4 macroblock rows is swept by a small team of
// specialised waves (compute waves that only touch LDS, a mover wave for memory); bands
// hand rows on through LDS inside a workgroup and through a write-through buffer between
// the workgroups (CUs) that share one unit (stream, slice). Every wait is a bounded spin.
// Frames overlap too: progress words in memory let the next frame's analysis follow this
// frame's deblocking.
//
// The reference delegates both stages to silicon (kernel/cedar.c:1135-1153
// programs REC/REF/MBINFO addresses, :1176 triggers); deblocking policy is the
// slice header's (cedar.c:1024-1029: idc 0, offsets 0).
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "enc_device.h"
#include "enc_wf_common.h"

namespace cedar {



// 4x4 Hadamard sign H[i][a] for rows (1,1,1,1) (1,1,-1,-1) (1,-1,-1,1) (1,-1,1,-1)
DEVI int hsign(int i, int a)
{
	// bit a of nibble i set = minus. row0 ++++ (0000), row1 ++-- (1100),
	// row2 +--+ (0110), row3 +-+- (1010)
	const uint32_t neg = (0xAu << 12) | (0x6u << 8) | (0xCu << 4) | 0x0u;
	return ((neg >> (4 * i + a)) & 1) ? -1 : 1;
}

// ------------------------------------------------------------------------
// Intra reconstruction tiles (k_intra_recon2, further down): each band of 4 macroblock rows
// is swept left to right, lane group g = lane>>4 works on row band+g, two macroblocks behind
// group g-1 (the top-right dependency), so all 64 lanes are busy.
//   Intra 4x4:   the 16 lanes of a group are the 16 samples of the current 4x4
//                block; the block loop is the serial chain the standard imposes.
//                Transforms run across lanes with DPP (quad_perm / row_mirror),
//                no LDS round trip: forward leaves coefficient (P[i],P[j]),
//                P = {0,2,1,3}, on lane (i,j), which is what the inverse wants.
//   Intra 16x16: the 16 lanes are the 16 4x4 blocks, each transformed in
//                registers; DC Hadamard through a 16-word LDS scratch.
//   chroma:      lanes 0..7 of the group = (plane, 4x4 block).
// ------------------------------------------------------------------------
#define IT_STRIDE 32 // luma tile: sample (x,y) at [(1+y)*32 + 8 + x], x in [-1,19], y in [-1,15]
#define IC_STRIDE 32 // chroma tile (nv12): sample (x,y,c) at [(1+y)*32 + 8 + 2x + c], x,y in [-1,7]

constexpr int DB2_PAIRS_MAX = WF_WAVES / 2;


// Everything a deblock workgroup keeps in LDS. It lives in DYNAMIC shared memory so that the
// luma and the chroma instantiation can sit in one kernel (k_deblock2_planes: one launch, the
// plane picked per workgroup) without their footprints adding up.
// PAIRS filter/mover wave pairs per workgroup: 8 (a whole CU: one unit's wavefront as fast as it goes)
// or 2 (a workgroup small enough to find room beside the analysis of other units)
template <int PL, int PAIRS>
struct Db2Shared {
	Db2GroupLds<PL> g[PAIRS * 4];
	// bottom rows of a band's last macroblock row, for the pair that filters the band below
	// (a pair with few partners is back at its ring while the band below its previous one still reads
	// it: two rings, taken in turn)
	uint8_t xbot[PAIRS][PAIRS < 8 ? 2 : 1][DB2_XRING][(PL & 1) ? 64 : 16];
	uint8_t xbotc[PAIRS][PAIRS < 8 ? 2 : 1][DB2_XRING][(PL & 2) ? 32 : 16];
	int staged[PAIRS], filtered[PAIRS], vdone[PAIRS]; // steps staged / filtered / through their vertical edges
	int ticket;
	int abort; // a wait of this workgroup ran out: stop waiting
	uint64_t qdb[52]; // MODE 3: per QP alpha, beta, tc0 of bS 1..3, QP_C (one byte each), from the session's table
};

extern __shared__ __attribute__((aligned(16))) uint8_t dyn_lds[];

// MODE 0: a P picture of inter macroblocks; 1: an IDR picture (all intra: bS 4 on macroblock edges, 3 inside);
// 2: a P picture with intra macroblocks in it (cyclic intra refresh): the strengths of both, per edge;
// 3: any picture of a session with regions of interest: as 2 (an IDR picture is the case "all intra"), and the
//    thresholds alpha / beta / tc0 are the macroblock's own, per edge class (left, top, inner): the mover
//    follows QP_Y along the row from the records and qp_t
template <int MODE, int PL, int PAIRS>
DEVI void deblock2_body(const Params &p, const int wg)
{
	// (4, 5: modes 0 and 1 for the pictures that may hold I_PCM macroblocks, Params::pcm -- kernels of their own,
	// so that the two extra words per macroblock the mover then loads cost the common pictures no register)
	constexpr bool PCM = MODE >= 4;
	constexpr int BASE = PCM ? MODE - 4 : MODE;
	constexpr bool IDR = BASE == 1, MIXED = BASE == 2 || BASE == 3, QPM = BASE == 3;
	constexpr bool DO_Y = (PL & 1) != 0, DO_C = (PL & 2) != 0;

	Db2Shared<PL, PAIRS> &S = *reinterpret_cast<Db2Shared<PL, PAIRS> *>(dyn_lds);
	Db2GroupLds<PL> *s_g = S.g;
	auto &s_xbot = S.xbot;
	auto &s_xbotc = S.xbotc;
	int *s_staged = S.staged, *s_filtered = S.filtered, *s_vdone = S.vdone;
	int &s_ticket = S.ticket;

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	// waves land on SIMD wave%4: filter k on SIMD k%4, its mover on SIMD (k+2)%4, so the
	// busy filter waves of a workgroup (bands come in pair order) never share a SIMD early
	const bool is_mover = wave >= PAIRS;
	const int pair = is_mover ? (wave + 6) & (PAIRS - 1) : wave;
	const int g = lane >> 4, l = lane & 15;
	const int NWG = p.wf_split;
	uint32_t *sync = p.wf_sync + WF_SYNC_WORDS * (PL == 2 ? 2 : 1);
	if (threadIdx.x == 0) {
		s_ticket = NWG > 1 ? (int)atomicAdd(&sync[0], 1u) : wg;
		S.abort = 0;
	}
	if (QPM && threadIdx.x < 52) {
		const QpRow &r = p.qtab[threadIdx.x];
		S.qdb[threadIdx.x] = (uint64_t)(uint32_t)r.alpha | (uint64_t)(uint32_t)r.beta << 8 | (uint64_t)(uint32_t)r.tc0[0] << 16 |
				     (uint64_t)(uint32_t)r.tc0[1] << 24 | (uint64_t)(uint32_t)r.tc0[2] << 32 | (uint64_t)(uint32_t)r.qpc << 40;
	}
	if (threadIdx.x < PAIRS) {
		s_staged[threadIdx.x] = 0;
		s_filtered[threadIdx.x] = 0;
		s_vdone[threadIdx.x] = 0;
	}

	__syncthreads();
	// a ticket beyond the launch means the sync words were not zero when the kernel started: refuse
	// to index memory with it (code 32 in the frame's error word) rather than fault
	if (s_ticket < 0 || s_ticket >= p.n_streams * p.slice_num * NWG) {
		if (threadIdx.x == 0)
			__hip_atomic_store(p.wf_err, 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		return;
	}
	const int unit = s_ticket / NWG, wj = s_ticket % NWG;
	const int st = unit / p.slice_num, sl = p.slice_first + unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, sl, &r0, &r1);

	Db2GroupLds<PL> &L = s_g[pair * 4 + g];
	const Db2GroupLds<PL> &LU = s_g[pair * 4 + (g > 0 ? g - 1 : 0)];
	uint8_t *rec_y = p.rec_y + (size_t)st * p.plane_y_stride;
	uint8_t *rec_uv = p.rec_uv + (size_t)st * p.plane_uv_stride;
	const int nbands = (r1 - r0 + 3) >> 2;
	const int b_lo = nbands * wj / NWG, b_hi = nbands * (wj + 1) / NWG;
	// Rows of a band follow each other ONE macroblock apart: filtering (x, y) needs (x+1, y-1)
	// only for the three columns its first vertical edge changes in (x, y-1), and only from the
	// horizontal edges on. All groups do their vertical edges, then fetch the rows above (with
	// that patch, fresh from the group above), then do their horizontal edges.
	const int nsteps = p.mb_w + 3;
	// edge code by boundary strength 0..3, one byte each (bS 4 = 0x80: the strong filter has no tc0)
	const uint32_t cod_y = (0x80u | p.tc0_y[0]) << 8 | (0x80u | p.tc0_y[1]) << 16 | (0x80u | p.tc0_y[2]) << 24;
	const uint32_t cod_c = (0x80u | p.tc0_c[0]) << 8 | (0x80u | p.tc0_c[1]) << 16 | (0x80u | p.tc0_c[2]) << 24;
	uint32_t *xprog = sync + 16 + unit * WF_MAX_SPLIT;
	uint32_t *xabort = sync + 1, *xerr = p.wf_err;
	const WaitCtx xw{xerr, &S.abort, xabort};
	uint64_t *hb_out = p.wf_hbuf + ((size_t)unit * p.wf_hb_slots + wj) * p.mb_w * WF_HBUF_GRANULES;
	const uint64_t *hb_in = p.wf_hbuf + ((size_t)unit * p.wf_hb_slots + (wj > 0 ? wj - 1 : 0)) * p.mb_w * WF_HBUF_GRANULES;
	uint32_t xseen = 0;

	int iter = 0;
	for (int band = b_lo + pair; band < b_hi; band += PAIRS, iter++) {
		const int base = iter * (nsteps + 2); // absolute step numbering of this pair's counters
		const int my = r0 + 4 * band + g;
		const bool row_ok = my < r1;
		const bool top_edge = my > r0;
		// bands of one workgroup hand their bottom rows on through LDS (s_xbot), filter wave to
		// filter wave: the pair below follows two macroblocks behind like a group of the same
		// wave would; its movers stay out of that dependency chain
		const int rel = band - b_lo;
		const bool in_top = rel > 0, in_bot = band + 1 < b_hi;
		const int pair_up = (rel - 1) & (PAIRS - 1), base_up = ((rel - 1) / PAIRS) * (nsteps + 2);
		const int pair_dn = (rel + 1) & (PAIRS - 1), base_dn = ((rel + 1) / PAIRS) * (nsteps + 2);
		constexpr int NBUF = PAIRS < 8 ? 2 : 1;
		const int wb = iter & (NBUF - 1), rb = ((rel - 1) / PAIRS) & (NBUF - 1); // this band's ring, the ring of the band above
		const bool feeds_next = g == 3 && band + 1 < nbands; // its last 4 rows are stored by whoever filters below
		const bool xwg_band = band == b_lo && b_lo > 0;
		const bool xwg_top = g == 0 && xwg_band;
		const bool is_xprod = g == 3 && band == b_hi - 1 && b_hi < nbands;

		if (!is_mover) {
			// =========================== filter wave ===========================
			// The hand-off ring of this pair may still hold the end of the pair's PREVIOUS band (8
			// bands up), which the band below that one reads: wait until it is through before the
			// ring is written again (only when a workgroup has more bands than pairs: NBUF times more).
			if (iter >= NBUF && in_bot) {
				const int relc = rel - NBUF * PAIRS + 1;
				lds_wait_ge(&s_filtered[relc & (PAIRS - 1)], (relc / PAIRS) * (nsteps + 2) + nsteps, xw);
			}
			for (int t = 0; t < nsteps; t++) {
				const int mx = t - g;
				const bool act = row_ok && mx >= 0 && mx < p.mb_w;
				const bool left_edge = mx > 0;
				const int slot = t % 3, pslot = (t + 2) % 3;
				lds_wait_ge(&s_staged[pair], base + t + 1, xw);
				if (act) {
					if (left_edge) { // previous tile's last 4 columns (final) become the left neighbours
						if (DO_Y)
							*(uint32_t *)(L.y[slot] + (4 + l) * DT_STRIDE + 12) = *(const uint32_t *)(L.y[pslot] + (4 + l) * DT_STRIDE + 28);
						if (DO_C && l < 8)
							*(uint32_t *)(L.c[slot] + (2 + l) * DC_STRIDE + 12) = *(const uint32_t *)(L.c[pslot] + (2 + l) * DC_STRIDE + 28);
					}
				}
				WAVE_SYNC();
				const uint32_t wv = DO_Y ? L.bsv[slot][l] : 0u, wh = DO_Y ? L.bsh[slot][l] : 0u, wc = DO_C ? L.bsc[slot][l] : 0u;
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
				const bool any_work = __any((wv | wh | wc) != 0); // codes are 0 on idle lanes
				if (act && any_work) {
					// ---- vertical edges: lane = sample row l ----
					if (DO_Y) {
						int px[20];
						const uint32_t *rowp = (const uint32_t *)(L.y[slot] + (4 + l) * DT_STRIDE + 12);
#pragma unroll
						for (int w = 0; w < 5; w++) {
							const uint32_t v = rowp[w];
							px[4 * w] = v & 0xff;
							px[4 * w + 1] = (v >> 8) & 0xff;
							px[4 * w + 2] = (v >> 16) & 0xff;
							px[4 * w + 3] = v >> 24;
						}
#pragma unroll
						for (int e = 0; e < 4; e++) {
							const int q = 4 + 4 * e;
							if (!IDR && !__any((wv >> (8 * e)) & 0x80u))
								continue;
							if (IDR && e == 0) {
								filt_luma_s2(wv, e, ayL, byL, px[q - 4], px[q - 3], px[q - 2], px[q - 1],
									     px[q], px[q + 1], px[q + 2], px[q + 3]);
							} else if (MIXED && e == 0) { // code 0xC0 = bS 4: one of the two filters per lane
								const uint32_t ws = strong_part(wv, 0), wn = wv ^ ws;
								if (__any(ws))
									filt_luma_s2(ws, e, ayL, byL, px[q - 4], px[q - 3], px[q - 2], px[q - 1],
										     px[q], px[q + 1], px[q + 2], px[q + 3]);
								if (__any(wn & 0x80u))
									filt_luma_n2(wn, e, -ayL, -byL,
										     px[q - 3], px[q - 2], px[q - 1], px[q], px[q + 1], px[q + 2]);
							} else {
								filt_luma_n2(wv, e, e == 0 ? -ayL : -ayI, e == 0 ? -byL : -byI,
									     px[q - 3], px[q - 2], px[q - 1], px[q], px[q + 1], px[q + 2]);
							}
						}
						uint32_t *roww = (uint32_t *)(L.y[slot] + (4 + l) * DT_STRIDE + 12);
#pragma unroll
						for (int w = 0; w < 5; w++)
							roww[w] = (uint32_t)px[4 * w] | ((uint32_t)px[4 * w + 1] << 8) | ((uint32_t)px[4 * w + 2] << 16) | ((uint32_t)px[4 * w + 3] << 24);
					}
					if (DO_C) { // lane = (plane c = l>>3, chroma row r = l&7); edges at chroma x = 0 and 4
						const int c = l >> 3, r = l & 7;
						uint8_t *cp = L.c[slot] + (2 + r) * DC_STRIDE + 16 + c;
						int a1 = cp[-4], a0 = cp[-2], b0 = cp[0], b1 = cp[2];
						int c1 = cp[4], c0 = cp[6], d0 = cp[8], d1 = cp[10];
						if (IDR) {
							filt_chroma_s2(wc, 0, -acL, -bcL, a1, a0, b0, b1);
						} else if (MIXED) {
							const uint32_t ws = strong_part(wc, 0);
							filt_chroma_s2(ws, 0, -acL, -bcL, a1, a0, b0, b1);
							filt_chroma_n2(wc ^ ws, 0, -acL, -bcL, a1, a0, b0, b1);
						} else {
							filt_chroma_n2(wc, 0, -acL, -bcL, a1, a0, b0, b1);
						}
						filt_chroma_n2(wc, 1, -acI, -bcI, c1, c0, d0, d1);
						cp[-2] = (uint8_t)a0;
						cp[0] = (uint8_t)b0;
						cp[6] = (uint8_t)c0;
						cp[8] = (uint8_t)d0;
					}
				}
				WAVE_SYNC();
				// columns 12..15 of the macroblock to the left are final now (this macroblock's first
				// vertical edge was the last thing to touch them): the band below wants them before ITS
				// horizontal edges of this step
				if (act && left_edge && g == 3 && in_bot) {
					if (DO_Y && l >= 12)
						*(uint32_t *)(s_xbot[pair][wb][(mx - 1) & (DB2_XRING - 1)] + (l - 12) * 16 + 12) = *(const uint32_t *)(L.y[slot] + (4 + l) * DT_STRIDE + 12);
					else if (DO_C && l >= 6 && l < 8)
						*(uint32_t *)(s_xbotc[pair][wb][(mx - 1) & (DB2_XRING - 1)] + (l - 6) * 16 + 12) = *(const uint32_t *)(L.c[slot] + (2 + l) * DC_STRIDE + 12);
				}
				lds_signal(&s_vdone[pair], base + t + 1, lane);
				// rows above: saved by the group above a step ago, their last columns patched by its
				// vertical edges of this step; group 0 takes them from the pair above (or had them
				// staged by the mover when they come from another workgroup)
				if (in_top && t < p.mb_w) {
					if (t + 1 < p.mb_w)
						lds_wait_ge(&s_vdone[pair_up], base_up + t + 5, xw);    // (t+1, above) through its vertical edges
					else
						lds_wait_ge(&s_filtered[pair_up], base_up + t + 4, xw); // (t, above) done, nothing to its right
				}
				if (act && top_edge) {
					if (g > 0) {
						const bool patched = mx + 1 < p.mb_w;
						if (DO_Y && l < 4) {
							u32x4 v = *(const u32x4 *)(LU.bot[mx & 3] + l * 16);
							if (patched)
								v.w = *(const uint32_t *)(LU.y[slot] + (16 + l) * DT_STRIDE + 12);
							*(u32x4 *)(L.y[slot] + l * DT_STRIDE + 16) = v;
						} else if (DO_C && l >= 4 && l < 6) {
							u32x4 v = *(const u32x4 *)(LU.botc[mx & 3] + (l - 4) * 16);
							if (patched)
								v.w = *(const uint32_t *)(LU.c[slot] + (4 + l) * DC_STRIDE + 12);
							*(u32x4 *)(L.c[slot] + (l - 4) * DC_STRIDE + 16) = v;
						}
					} else if (in_top) {
						if (DO_Y && l < 4)
							*(u32x4 *)(L.y[slot] + l * DT_STRIDE + 16) = *(const u32x4 *)(s_xbot[pair_up][rb][mx & (DB2_XRING - 1)] + l * 16);
						else if (DO_C && l >= 4 && l < 6)
							*(u32x4 *)(L.c[slot] + (l - 4) * DC_STRIDE + 16) = *(const u32x4 *)(s_xbotc[pair_up][rb][mx & (DB2_XRING - 1)] + (l - 4) * 16);
					}
				}
				WAVE_SYNC();
				if (act && any_work) {
					// ---- horizontal edges: lane = sample column l ----
					if (DO_Y) {
						int py[20];
						uint8_t *colp = L.y[slot] + 16 + l;
#pragma unroll
						for (int r = 0; r < 20; r++)
							py[r] = colp[r * DT_STRIDE];
#pragma unroll
						for (int e = 0; e < 4; e++) {
							const int q = 4 + 4 * e;
							if (!IDR && !__any((wh >> (8 * e)) & 0x80u))
								continue;
							if (IDR && e == 0) {
								filt_luma_s2(wh, e, ayT, byT, py[q - 4], py[q - 3], py[q - 2], py[q - 1],
									     py[q], py[q + 1], py[q + 2], py[q + 3]);
							} else if (MIXED && e == 0) {
								const uint32_t ws = strong_part(wh, 0), wn = wh ^ ws;
								if (__any(ws))
									filt_luma_s2(ws, e, ayT, byT, py[q - 4], py[q - 3], py[q - 2], py[q - 1],
										     py[q], py[q + 1], py[q + 2], py[q + 3]);
								if (__any(wn & 0x80u))
									filt_luma_n2(wn, e, -ayT, -byT,
										     py[q - 3], py[q - 2], py[q - 1], py[q], py[q + 1], py[q + 2]);
							} else {
								filt_luma_n2(wh, e, e == 0 ? -ayT : -ayI, e == 0 ? -byT : -byI,
									     py[q - 3], py[q - 2], py[q - 1], py[q], py[q + 1], py[q + 2]);
							}
						}
#pragma unroll
						for (int r = 1; r < 19; r++)
							colp[r * DT_STRIDE] = (uint8_t)py[r];
					}
					if (DO_C) { // lane = byte column l of the nv12 row
						uint8_t *cp = L.c[slot] + 16 + l;
						int a1 = cp[0], a0 = cp[DC_STRIDE], b0 = cp[2 * DC_STRIDE], b1 = cp[3 * DC_STRIDE];
						int c1 = cp[4 * DC_STRIDE], c0 = cp[5 * DC_STRIDE], d0 = cp[6 * DC_STRIDE], d1 = cp[7 * DC_STRIDE];
						if (IDR) {
							filt_chroma_s2(wc, 2, -acT, -bcT, a1, a0, b0, b1);
						} else if (MIXED) {
							const uint32_t ws = strong_part(wc, 2);
							filt_chroma_s2(ws, 2, -acT, -bcT, a1, a0, b0, b1);
							filt_chroma_n2(wc ^ ws, 2, -acT, -bcT, a1, a0, b0, b1);
						} else {
							filt_chroma_n2(wc, 2, -acT, -bcT, a1, a0, b0, b1);
						}
						filt_chroma_n2(wc, 3, -acI, -bcI, c1, c0, d0, d1);
						cp[DC_STRIDE] = (uint8_t)a0;
						cp[2 * DC_STRIDE] = (uint8_t)b0;
						cp[5 * DC_STRIDE] = (uint8_t)c0;
						cp[6 * DC_STRIDE] = (uint8_t)d0;
					}
				}
				WAVE_SYNC();
				if (in_bot && t - 3 >= DB2_XRING && t - 3 < p.mb_w) // ring slot of group 3: consumed by the pair below?
					lds_wait_ge(&s_filtered[pair_dn], base_dn + (t - 3 - DB2_XRING) + 1, xw);
				if (act) { // bottom rows for the row below; columns 12..15 of the previous MB are final now
					if (DO_Y && l >= 12) {
						const u32x4 rowv = *(const u32x4 *)(L.y[slot] + (4 + l) * DT_STRIDE + 16);
						const uint32_t patch = *(const uint32_t *)(L.y[slot] + (4 + l) * DT_STRIDE + 12);
						*(u32x4 *)(L.bot[mx & 3] + (l - 12) * 16) = rowv;
						if (left_edge)
							*(uint32_t *)(L.bot[(mx - 1) & 3] + (l - 12) * 16 + 12) = patch;
						if (g == 3 && in_bot)
							*(u32x4 *)(s_xbot[pair][wb][mx & (DB2_XRING - 1)] + (l - 12) * 16) = rowv;
					} else if (DO_C && l >= 6 && l < 8) {
						const u32x4 rowv = *(const u32x4 *)(L.c[slot] + (2 + l) * DC_STRIDE + 16);
						const uint32_t patch = *(const uint32_t *)(L.c[slot] + (2 + l) * DC_STRIDE + 12);
						*(u32x4 *)(L.botc[mx & 3] + (l - 6) * 16) = rowv;
						if (left_edge)
							*(uint32_t *)(L.botc[(mx - 1) & 3] + (l - 6) * 16 + 12) = patch;
						if (g == 3 && in_bot)
							*(u32x4 *)(s_xbotc[pair][wb][mx & (DB2_XRING - 1)] + (l - 6) * 16) = rowv;
					}
				}
				lds_signal(&s_filtered[pair], base + t + 1, lane);
			}
			continue;
		}

		// ============================== mover wave ==============================
		// Frame pipelining: this frame's analysis runs ahead of us on another stream in diagonal
		// stripes (prog_inter), and the next frame's analysis follows OUR write-back (prog_db).
		// Two ways of counting progress. This kernel sweeps with ONE macroblock of lag per row: at step t
		// the four rows of the band are at x + y = t + y0 (y0 = the band's first row), and "everything
		// with x + y < E is through" says exactly how far it is -- that is what it publishes and what a
		// P picture's analysis (stripes of x + y, published by k_gate) is asked for. The intra wavefront
		// needs TWO macroblocks of lag per row (x + 2*y) and publishes in that measure: an IDR picture's
		// deblocking waits for it in x + 2*y, where the band's rows are at t + 2*y0 .. t + 2*y0 + 3.
		// (Counting everything in x + 2*y is true too, but it asks the analysis of a P picture to be 3
		// diagonals per band AHEAD of where the band is -- 52 for the bottom band of a 1080p picture.)
		const int y0b = r0 + 4 * band;
		AnalysisWait aw{p.prog_inter ? 0u : 0x7fffffffu, IDR || (QPM && p.is_idr), y0b, unit, band, nbands, PL}; // (mode 3 serves the IDR pictures of its sessions too)
		uint32_t *const my_prog = p.prog_db ? p.prog_db + ((PL == 2 ? p.n_streams * p.slice_num : 0) + unit) * PROG_BANDS + (band < PROG_BANDS ? band : PROG_BANDS - 1) : nullptr;
		auto publish_diag = [&](uint32_t d) { // every macroblock of this band with x + 2*y < d is written back
			if (!my_prog)
				return;
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			if (lane == 0) {
				__hip_atomic_store(my_prog, p.prog_tag | d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				if (PL == 3)
					__hip_atomic_store(my_prog + p.n_streams * p.slice_num * PROG_BANDS, p.prog_tag | d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			}
		};
		const size_t row_mb = (size_t)st * p.n_mb + (size_t)(row_ok ? my : r0) * p.mb_w;
		// two register sets: the loads of step t+3 are issued while step t+1 is staged (memory
		// latency here is about one whole step)
		struct MoverRegs {
			u32x4 in, c;
			uint64_t topx;
			uint32_t mv, mvt; // motion-vector dword of MbInfo, kept opaque until staged
			uint32_t ty, tyt; // MIXED: its first dword (type in the low byte)
			uint8_t qc, qt;   // MODE 3: QP the macroblock, and the one above it, is quantised with (qp_t)
			uint8_t nz, nzt;
			bool top_ok;
		};
		MoverRegs RA{}, RB{};
		EdgeCarry carry; // of the macroblock staged last (the left neighbour of the next)
		int pending_x = -1;

		auto load_step = [&](int t, bool block, MoverRegs &R) { // global loads for step t into the r_* registers
			const int nmx = t - g, mx0 = t;
			const bool nact = row_ok && nmx >= 0 && nmx < p.mb_w;
			const int cmx = nact ? nmx : 0;
			const size_t a = row_mb + cmx;
			const int x0 = cmx * 16, y0 = (row_ok ? my : r0) * 16;
			if (t < nsteps)
				wait_analysis(p, aw, t, xerr);
			// readiness of the rows above first: its acquire must not sit behind this step's own loads
			R.top_ok = true;
			if (xwg_band && mx0 < p.mb_w) { // rows above live in another workgroup

				const uint32_t need = (uint32_t)mx0 + 1u;
				if (xseen < need)
					xseen = xload32(&xprog[wj - 1]);
				if (xseen < need && block)
					xseen = xwait(&xprog[wj - 1], xabort, xerr, need);
				R.top_ok = xseen >= need;
			}
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
			if (xwg_top && R.top_ok && nact && ((DO_Y && l < 8) || (DO_C && l >= 8 && l < WF_HBUF_GRANULES)))
				R.topx = xload64(hb_in + (size_t)cmx * WF_HBUF_GRANULES + l);
		};

		auto stage_step = [&](int t, const MoverRegs &R) { // r_* registers -> tile slot t%3 + packed boundary strengths
			const int mx = t - g;
			const bool act = row_ok && mx >= 0 && mx < p.mb_w;
			const bool left_edge = mx > 0;
			const int slot = t % 3;
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
				if (xwg_top) { // rows above from the other workgroup (inside a workgroup the filter wave fetches them)
					if (DO_Y && l < 8)
						*(uint64_t *)(L.y[slot] + (l >> 1) * DT_STRIDE + 16 + (l & 1) * 8) = R.topx;
					else if (DO_C && l >= 8 && l < 12)
						*(uint64_t *)(L.c[slot] + ((l - 8) >> 1) * DC_STRIDE + 16 + (l & 1) * 8) = R.topx;
				}
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

		auto writeback_step = [&](int t) { // filtered tile of step t -> memory, next workgroup
			const int mx = t - g;
			const bool act = row_ok && mx >= 0 && mx < p.mb_w;
			const bool left_edge = mx > 0;
			const int slot = t % 3;
			const int x0 = mx * 16, y0 = my * 16;
			if (!act)
				return;
			if (DO_Y && !(feeds_next && l >= 12)) {
				*(u32x4 *)(rec_y + (size_t)(y0 + l) * p.cw + x0) = *(const u32x4 *)(L.y[slot] + (4 + l) * DT_STRIDE + 16);
				if (left_edge)
					*(uint32_t *)(rec_y + (size_t)(y0 + l) * p.cw + x0 - 4) = *(const uint32_t *)(L.y[slot] + (4 + l) * DT_STRIDE + 12);
			}
			if (DO_C && l < 8 && !(feeds_next && l >= 6)) {
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
			if (is_xprod) {
				const bool my_gran = (DO_Y && l < 8) || (DO_C && l >= 8 && l < WF_HBUF_GRANULES);
				if (left_edge && my_gran) {
					const uint8_t *src = l < 8 ? L.bot[(mx - 1) & 3] + (l >> 1) * 16 + (l & 1) * 8
								   : L.botc[(mx - 1) & 3] + ((l - 8) >> 1) * 16 + (l & 1) * 8;
					xstore64(hb_out + (size_t)(mx - 1) * WF_HBUF_GRANULES + l, *(const uint64_t *)src);
				}
				if (mx == p.mb_w - 1 && my_gran) {
					const uint8_t *src = l < 8 ? L.bot[mx & 3] + (l >> 1) * 16 + (l & 1) * 8
								   : L.botc[mx & 3] + ((l - 8) >> 1) * 16 + (l & 1) * 8;
					xstore64(hb_out + (size_t)mx * WF_HBUF_GRANULES + l, *(const uint64_t *)src);
				}
				pending_x = mx == p.mb_w - 1 ? p.mb_w : mx;
			}
		};

		auto publish = [&]() { // everything issued an iteration ago has completed by now
			if (b_hi < nbands && band == b_hi - 1) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if (is_xprod && l == 0 && pending_x >= 0)
					xstore32(&xprog[wj], (uint32_t)pending_x);
				pending_x = -1;
			}
		};

		auto mover_iter = [&](int t, MoverRegs &R) { // R holds step t+1 and is refilled with step t+3
			if (t >= 1)
				lds_wait_ge(&s_filtered[pair], base + t, xw); // step t-1 filtered
			publish();
			if ((t & 7) == 0 && t >= 8) // steps <= t-2 were written back an iteration ago
				publish_diag((uint32_t)(t - 1 + y0b)); // every macroblock of the band with x + y < that
			if (t + 1 < nsteps) {
				if (!R.top_ok)
					load_step(t + 1, true, R);
				stage_step(t + 1, R);
				// fault injection (tests): from step 8 of the first band on, this mover stops reporting
				const bool lost = p.fault_inject == 1 && unit == 0 && band == 0 && t >= 8;
				lds_signal(&s_staged[pair], lost ? base + 1 : base + t + 2, lane);
				load_step(t + 3, false, R); // may wait for this frame's analysis: after the filter wave has its tile
			}
			if (t >= 1)
				writeback_step(t - 1);
		};
		// prologue: step 0 staged, loads of steps 1 and 2 in flight
		load_step(0, true, RA);
		load_step(1, false, RB);
		stage_step(0, RA);
		load_step(2, false, RA);
		lds_signal(&s_staged[pair], base + 1, lane);
		for (int t = 0; t <= nsteps; t += 2) {
			mover_iter(t, RB);
			if (t + 1 <= nsteps)
				mover_iter(t + 1, RA);
		}
		publish();
		publish_diag((uint32_t)PROG_ALL);
	}
}

// planes: 1 luma, 2 chroma, 3 both planes in every wave pair (many units), 4 both planes in
// one launch as separate workgroups (first half of the grid luma, second half chroma)
template <int MODE, int PAIRS>
__global__ __launch_bounds__(128 * PAIRS) void k_deblock2_planes(Params p, int planes)
{
	const int nwg = p.n_streams * p.slice_num * p.wf_split;
	if (planes == 4) {
		if ((int)blockIdx.x < nwg)
			deblock2_body<MODE, 1, PAIRS>(p, (int)blockIdx.x);
		else
			deblock2_body<MODE, 2, PAIRS>(p, (int)blockIdx.x - nwg);
	} else if (planes == 1) {
		deblock2_body<MODE, 1, PAIRS>(p, (int)blockIdx.x);
	} else if (planes == 2) {
		deblock2_body<MODE, 2, PAIRS>(p, (int)blockIdx.x);
	} else {
		deblock2_body<MODE, 3, PAIRS>(p, (int)blockIdx.x);
	}
	if (!p.db_self_reset)
		return;
	// Epilogue of a pipelined frame: the last workgroup to get here leaves the sync words of both
	// deblock regions zeroed for the next frame that uses this slot (nobody reads them any more:
	// every workgroup has counted itself out) and tells the host that the frame is deblocked.
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t *r1 = p.wf_sync + WF_SYNC_WORDS, *r2 = p.wf_sync + 2 * WF_SYNC_WORDS;
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
		if (__hip_atomic_fetch_add(&r1[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			const int used = 16 + p.n_streams * p.slice_num * WF_MAX_SPLIT;
			for (int i = 0; i < used; i++) {
				xstore32(&r1[i], 0u);
				xstore32(&r2[i], 0u);
			}
			if (p.done_flag)
				__hip_atomic_store(p.done_flag, p.done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

// ------------------------------------------------------------------------
// k_intra_recon2: the intra wavefront with specialised waves, as k_deblock2. A step of
// the single-wave kernel cost ~37000 cycles: ~3700 waiting for its own loads, 20500
// Intra4x4 chain, 1700 Intra16x16, 10000 chroma, ~1000 write-back and fences. Chroma
// prediction only needs chroma neighbours and the loads / stores need nobody, so a band
// is served by a TRIPLE of waves:
//   luma wave    Intra4x4 / Intra16x16 chain on its LDS tile, levels and nnz to memory
//   chroma wave  the chroma blocks on its own tile (same step, other SIMD)
//   mover wave   loads of step t+3 in flight, stages step t+1 (source samples, record,
//                packed 4x4 modes, rows above for group 0); writes step t-1 back
// The step is then the luma chain alone. Bottom rows travel between the groups of one
// compute wave through LDS rings (boty / botc); band and workgroup boundaries as before.
// ------------------------------------------------------------------------
// IR_TRIPLES (enc_internal.h) = 2: six waves, at most two on a SIMD, so each has 256 registers and nothing spills
// (four triples = 12 waves = 3 per SIMD had 168 each and kept 33 registers in scratch on the chain)

struct __attribute__((aligned(16))) Ir2GroupLds {
	uint8_t t[17 * IT_STRIDE];  // luma wave's tile (persists: column 15 is the next left neighbour)
	uint8_t c[9 * IC_STRIDE];   // chroma wave's tile
	uint8_t srcy[3][256];       // staged source, ring by step
	uint8_t srcc[3][128];
	uint8_t outy[3][256];       // finished reconstruction for the mover
	uint8_t outc[3][128];
	uint8_t boty[4][16];        // last luma row of macroblock mx (slot mx&3) for the group below
	uint8_t botc[4][16];        // last chroma row
	uint8_t topy[3][32];        // group 0 only, laid out as tile row 0: [7] x=-1, [8..23], [24..27] x=16..19
	uint8_t topc[3][32];        // [6..7] pair x=-1, [8..23]
	uint32_t info[3][4];        // [0..1] MbInfo, [2..3] the 16 Intra4x4 modes, one nibble each
	uint32_t qpw[4];            // regions of interest: QP_Y of the macroblock, ring by step
	int cbp_y[3], cbp_yb[3], cbp_c[3]; // luma bits found by the luma wave / by the chroma wave, chroma bits
	int dc[16], dcl[16], cdc[8], cdl[8];
};

// ---- one Intra4x4 block, lane = sample, shared by the two compute waves of a triple ----
struct I4Ctx {
	uint8_t *T;              // luma tile of the group
	const uint8_t *SY;       // staged source macroblock
	int16_t *cf_lane;        // this lane's level slot of block 0
	uint8_t *nnz_mb;
	uint64_t modes64;        // the 16 modes, one nibble each
	bool top_mb, left_mb, tr_mb, is4;
	int g, l, pi, pj, zz, my_mf, my_dv, f_intra, qbits, qshift;
};

// table entry (kind + three edge offsets) of block idx for this lane; neighbour availability
// (clause 6.4.11.4) as masks over the coding index: a constant inside the macroblock, the
// macroblock's own neighbours at its border
DEVI uint32_t i4_tab(const uint32_t *ptab, const I4Ctx &c, int idx)
{
	constexpr uint32_t kTrInside = 0x5744u;                             // 2,6,8,9,10,12,14
	constexpr uint32_t kTopRowNotLast = (1u << 0) | (1u << 1) | (1u << 4); // by == 0, bx < 3
	const uint32_t htr16 = kTrInside | (c.top_mb ? kTopRowNotLast : 0u) | (c.tr_mb ? 1u << 5 : 0u);
	const int mode = (int)((c.modes64 >> (4 * idx)) & 15u);
	return ptab[((((htr16 >> idx) & 1u) ? 9 : 0) + (mode < 9 ? mode : 2)) * 16 + c.l];
}

// predict, transform, quantise, reconstruct block idx; `tab` holds its table entry on entry
// and the entry of block `next` on return (fetched while this block is computed). Returns the
// block's bit of the coded block pattern.
DEVI int i4_block(const uint32_t *ptab, const I4Ctx &c, const LaneK &ci, const LaneK &cj, int idx, int next, uint32_t &tab)
{
	constexpr uint32_t kByGt0 = 0xffccu, kBxGt0 = 0xfafau; // blocks not in the first row / column
	const int bx = blk_x(idx), by = blk_y(idx);
	const uint32_t cur = tab;
	tab = i4_tab(ptab, c, next);
	int level = 0;
	if (c.is4) {
		const uint8_t *T = c.T;
		const int a0 = (1 + 4 * by) * IT_STRIDE + 8 + 4 * bx; // the block's first sample
		const int kind = (int)(cur >> 24);
		const int ea = T[a0 + (int)(int8_t)(cur & 0xff)], eb = T[a0 + (int)(int8_t)((cur >> 8) & 0xff)],
			  ec = T[a0 + (int)(int8_t)((cur >> 16) & 0xff)];
		const int st_ = (int)__builtin_amdgcn_sad_u8(*(const uint32_t *)(T + a0 - IT_STRIDE), 0u, 0u);
		const int sl_ = (int)T[a0 - 1] + (int)T[a0 + IT_STRIDE - 1] + (int)T[a0 + 2 * IT_STRIDE - 1] + (int)T[a0 + 3 * IT_STRIDE - 1];
		const bool ht = ((kByGt0 >> idx) & 1u) || c.top_mb, hl = ((kBxGt0 >> idx) & 1u) || c.left_mb;
		const int dcv = (ht && hl) ? (st_ + sl_ + 4) >> 3 : hl ? (sl_ + 2) >> 2 : ht ? (st_ + 2) >> 2 : 128;
		const int pv = sel(nz_mask(kind ^ 3),
				   sel(nz_mask(kind ^ 2), sel(nz_mask(kind ^ 1), eb, (eb + ec + 1) >> 1), (ea + 2 * eb + ec + 2) >> 2),
				   dcv);
		const int res = (int)c.SY[(4 * by + c.pi) * 16 + 4 * bx + c.pj] - pv;
		const int w = fdct_lanes(res, ci, cj);
		level = quant1(w, c.my_mf, c.f_intra, c.qbits);
		c.cf_lane[idx * 16] = (int16_t)level;
		const int d = (level * c.my_dv) << c.qshift;
		const int r = idct_lanes(d, ci, cj);
		c.T[a0 + c.pi * IT_STRIDE + c.pj] = (uint8_t)clip255(pv + r);
	}
	const uint64_t bal = __ballot(c.is4 && level != 0);
	const int nn = __popc((uint32_t)(bal >> (16 * c.g)) & 0xffffu);
	if (c.is4 && c.l == 0)
		c.nnz_mb[bx + 4 * by] = (uint8_t)nn;
	WAVE_SYNC();
	return (c.is4 && nn) ? (1 << (idx >> 2)) : 0;
}

template <bool ROI>
__global__ __launch_bounds__(64 * 3 * IR_TRIPLES) void k_intra_recon2(Params p)
{
	__shared__ QpRow s_qtab[ROI ? 52 : 1]; // regions of interest: the session's table of everything a QP decides
	__shared__ Ir2GroupLds s_g[IR_TRIPLES * 4];
	// last reconstructed row of every band's bottom macroblock row, for the triple that does the
	// band below (dynamic LDS: 2 x IR_TRIPLES rows of mb_w * 16 + 32 bytes; the rest of the
	// request is padding that keeps one workgroup per CU)
	extern __shared__ __attribute__((aligned(16))) uint8_t s_rows[];
	__shared__ int s_staged[IR_TRIPLES], s_done_y[IR_TRIPLES], s_done_c[IR_TRIPLES];
	// Intra4x4 stages finished in the current step by the luma wave (a) and the chroma wave (b),
	// as (absolute step) * 16 + stages: the two share the 16 blocks of a macroblock, see i4 below
	__shared__ int s_i4a[IR_TRIPLES], s_i4b[IR_TRIPLES];
	__shared__ int s_ticket, s_abort;
	// Intra4x4 predictors as tile offsets: entry [tr][mode][lane] = kind << 24 | the offsets (int8,
	// relative to the block's first sample in the tile) of edge samples E[k-1], E[k], E[k+1];
	// tr = 1 when the block's top-right neighbours exist, else E[t4..t7] fold onto t3
	__shared__ uint32_t s_ptab[2 * 9 * 16];

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	for (int i = threadIdx.x; i < 2 * 9 * 16; i += blockDim.x) {
		const int tkmax = i >= 9 * 16 ? 7 : 3;
		const int ent = p.pred4_tab[i % (9 * 16)];
		const int kind = ent >> 4, k0 = ent & 15;
		uint32_t w = (uint32_t)kind << 24;
#pragma unroll
		for (int j = 0; j < 3; j++) {
			const int k = max(1, min(13, k0 - 1 + j));
			// edge entry k -> tile position: E[1..4] = left column bottom-up, E[5] = top-left,
			// E[6..13] = top row t0..t7
			const int rel = (max(0, 5 - k) - 1) * IT_STRIDE + max(0, min(k - 6, tkmax) + 1) - 1;
			w |= (uint32_t)(rel & 0xff) << (8 * j);
		}
		s_ptab[i] = w;
	}
	// waves land on SIMD wave%4: the three waves of a triple sit on three different SIMDs
	static_assert(IR_TRIPLES == 2, "wave -> (role, triple) below is written for two triples on 4 SIMDs");
	// the luma waves (the long pole) alone on SIMDs 2 and 3; chroma 0 + mover 0 share SIMD 0, chroma 1 + mover 1 SIMD 1
	const int role = wave < 2 ? 1 : wave < 4 ? 0 : 2, trip = wave & 1; // 0 luma, 1 chroma, 2 mover
	const int g = lane >> 4, l = lane & 15;
	const int NWG = p.wf_split_intra;
	uint32_t *sync = p.wf_sync; // region 0
	if (threadIdx.x == 0) {
		s_ticket = NWG > 1 ? (int)atomicAdd(&sync[0], 1u) : (int)blockIdx.x;
		s_abort = 0;
	}
	if (threadIdx.x < IR_TRIPLES) {
		s_staged[threadIdx.x] = 0;
		s_done_y[threadIdx.x] = 0;
		s_done_c[threadIdx.x] = 0;
		s_i4a[threadIdx.x] = 0;
		s_i4b[threadIdx.x] = 0;
	}
	__syncthreads();
	// a ticket beyond the launch means the sync words were not zero when the kernel started: refuse
	// to index memory with it (code 32 in the frame's error word) rather than fault
	if (s_ticket < 0 || s_ticket >= p.n_streams * p.slice_num * NWG) {
		if (threadIdx.x == 0)
			__hip_atomic_store(p.wf_err, 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		return;
	}
	const int unit = s_ticket / NWG, wj = s_ticket % NWG;
	const int st = unit / p.slice_num, sl = p.slice_first + unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, sl, &r0, &r1);
	uint32_t *xprog = sync + 16 + unit * WF_MAX_SPLIT;
	uint32_t *xabort = sync + 1, *xerr = p.wf_err;
	const WaitCtx xw{xerr, &s_abort, xabort};
	// hand-off granules per macroblock: 0,1 = last luma row, 2,3 = last chroma row
	uint64_t *hb_out = p.wf_hbuf_intra + ((size_t)unit * p.wf_hb_slots + wj) * p.mb_w * WF_HBUF_GRANULES;
	const uint64_t *hb_in = p.wf_hbuf_intra + ((size_t)unit * p.wf_hb_slots + (wj > 0 ? wj - 1 : 0)) * p.mb_w * WF_HBUF_GRANULES;
	uint32_t xseen = 0;

	Ir2GroupLds &L = s_g[trip * 4 + g];
	const Ir2GroupLds &LU = s_g[trip * 4 + (g > 0 ? g - 1 : 0)];
	const uint8_t *src_y = src_luma(p, st);
	const uint8_t *src_uv = src_chroma(p, st);
	uint8_t *rec_y = p.rec_y + (size_t)st * p.plane_y_stride;
	uint8_t *rec_uv = p.rec_uv + (size_t)st * p.plane_uv_stride;
	int f_intra = (1 << p.qbits) / 3, f_intra_c = (1 << p.qbits_c) / 3;
	const int nbands = (r1 - r0 + 3) >> 2;
	const int b_lo = nbands * wj / NWG, b_hi = nbands * (wj + 1) / NWG;
	const int nsteps = p.mb_w + 6;

	// lane constants of the Intra4x4 path (lane = sample of a 4x4 block)
	const int pi = l >> 2, pj = l & 3;
	const int kv = pi == 1 ? 2 : pi == 2 ? 1 : pi, kh = pj == 1 ? 2 : pj == 2 ? 1 : pj;
	const int rpos = 4 * kv + kh; // raster coefficient position held by this lane
	const int pclass = kPosClass[rpos];
	int zz = 0;
#pragma unroll
	for (int z = 0; z < 16; z++)
		zz = kZigzag[z] == rpos ? z : zz;
	const LaneK ci = lane_consts(pi), cj = lane_consts(pj);
	int my_mf = pclass == 0 ? p.mf[0] : pclass == 1 ? p.mf[1] : p.mf[2];
	int my_dv = pclass == 0 ? p.dv[0] : pclass == 1 ? p.dv[1] : p.dv[2];
	int my_mfc = pclass == 0 ? p.mfc[0] : pclass == 1 ? p.mfc[1] : p.mfc[2];
	int my_dvc = pclass == 0 ? p.dvc[0] : pclass == 1 ? p.dvc[1] : p.dvc[2];
	// what the QP decides: the picture's (Params), or -- regions of interest -- the macroblock's, looked up
	// at the start of every step from the QP the mover staged with the macroblock
	int q_mf[3] = { p.mf[0], p.mf[1], p.mf[2] }, q_dv[3] = { p.dv[0], p.dv[1], p.dv[2] };
	int q_qbits = p.qbits, q_qshift = p.qshift, q_qbits_c = p.qbits_c, q_qshift_c = p.qshift_c, q_mfc0 = p.mfc[0], q_dvc0 = p.dvc[0];
	if (ROI) {
		for (int i = threadIdx.x; i < 52 * (int)(sizeof(QpRow) / 4); i += blockDim.x)
			((uint32_t *)s_qtab)[i] = ((const uint32_t *)p.qtab)[i];
		__syncthreads();
	}

	int iter = 0;
	for (int band = b_lo + trip; band < b_hi; band += IR_TRIPLES, iter++) {
		const int base = iter * (nsteps + 2);
		const int my = r0 + 4 * band + g;
		const bool row_ok = my < r1;
		const bool top_mb = my > r0;
		const bool xwg_band = band == b_lo && b_lo > 0;
		const bool xwg_top = g == 0 && xwg_band;
		const bool is_xprod = g == 3 && band == b_hi - 1 && b_hi < nbands;
		// bands of one workgroup hand their last row on through LDS, compute wave to compute wave
		// (as groups of one wave do), without a detour through memory and the mover
		const int rel = band - b_lo, row_bytes = p.mb_w * 16 + 32;
		const bool in_top = rel > 0, in_bot = band + 1 < b_hi;
		const int trip_up = (rel + IR_TRIPLES - 1) % IR_TRIPLES, base_up = ((rel - 1) / IR_TRIPLES) * (nsteps + 2);
		uint8_t *const rowy_mine = s_rows + (size_t)(2 * trip) * row_bytes, *const rowc_mine = rowy_mine + row_bytes;
		const uint8_t *const rowy_up = s_rows + (size_t)(2 * trip_up) * row_bytes, *const rowc_up = rowy_up + row_bytes;
		auto load_q = [&](int slot) { // the macroblock's QP, staged by the mover, picks the rows
			if (!ROI)
				return;
			const QpRow &Y = s_qtab[L.qpw[slot] & 63u];
			const QpRow &C = s_qtab[Y.qpc];
#pragma unroll
			for (int k = 0; k < 3; k++) {
				q_mf[k] = Y.mf[k];
				q_dv[k] = Y.dv[k];
			}
			q_qbits = Y.qbits; q_qshift = Y.qshift; q_qbits_c = C.qbits; q_qshift_c = C.qshift; q_mfc0 = C.mf[0]; q_dvc0 = C.dv[0];
			my_mf = Y.mf[pclass]; my_dv = Y.dv[pclass]; my_mfc = C.mf[pclass]; my_dvc = C.dv[pclass];
			f_intra = (1 << q_qbits) / 3;
			f_intra_c = (1 << q_qbits_c) / 3;
		};

		if (role == 0) {
			// ============================ luma wave ============================
			for (int t = 0; t < nsteps; t++) {
				const int mx = t - 2 * g;
				const bool act = row_ok && mx >= 0 && mx < p.mb_w;
				const bool left_mb = mx > 0, tr_mb = top_mb && mx < p.mb_w - 1;
				const int slot = t % 3;
				const size_t mbaddr = (size_t)st * p.n_mb + (size_t)(act ? my : r0) * p.mb_w + (act ? mx : 0);
				lds_wait_ge(&s_staged[trip], base + t + 1, xw);
				load_q(slot);
				const uint32_t miw = L.info[slot][0];
				const uint64_t modes64 = (uint64_t)L.info[slot][2] | (uint64_t)L.info[slot][3] << 32;
				const int mb_type = (int)(int8_t)(miw & 0xff), i16_mode = (int)((miw >> 8) & 0xff);
				const uint8_t *SY = L.srcy[slot];
				if (in_top && t < p.mb_w) // the macroblock above-right of group 0's is reconstructed
					lds_wait_ge(&s_done_y[trip_up], base_up + (t + 8 < nsteps ? t + 8 : nsteps), xw);
				if (act) {
					if (left_mb) // left neighbours = last column of the previous macroblock
						L.t[(1 + l) * IT_STRIDE + 7] = L.t[(1 + l) * IT_STRIDE + 23];
					if (top_mb) {
						if (g == 0 && in_top) { // the triple above left its last row in LDS
							const uint8_t *rp = rowy_up + 16 + mx * 16;
							if (l == 0)
								*(u32x4 *)(L.t + 8) = *(const u32x4 *)rp;
							else if (l == 1 && tr_mb)
								*(uint32_t *)(L.t + 24) = *(const uint32_t *)(rp + 16);
							else if (l == 2 && left_mb)
								L.t[7] = rp[-1];
						} else if (g == 0) { // another workgroup's: staged by the mover
							if (l == 0)
								*(u32x4 *)(L.t + 8) = *(const u32x4 *)(L.topy[slot] + 8);
							else if (l == 1 && tr_mb)
								*(uint32_t *)(L.t + 24) = *(const uint32_t *)(L.topy[slot] + 24);
							else if (l == 2 && left_mb)
								L.t[7] = L.topy[slot][7];
						} else { // the group above finished these two, one and three steps ago
							if (l == 0)
								*(u32x4 *)(L.t + 8) = *(const u32x4 *)LU.boty[mx & 3];
							else if (l == 1 && tr_mb)
								*(uint32_t *)(L.t + 24) = *(const uint32_t *)LU.boty[(mx + 1) & 3];
							else if (l == 2 && left_mb)
								L.t[7] = LU.boty[(mx - 1) & 3][15];
						}
					}
				}
				WAVE_SYNC();

				int16_t *cf = p.coef + mbaddr * COEF_PER_MB;
				int16_t *cf_lane = cf + COEF_LUMA + zz;
				uint8_t *nnz_mb = p.nnz + mbaddr * NNZ_PER_MB;
				const bool is4 = act && mb_type == MB_I4x4, is16 = act && mb_type == MB_I16x16;
				int cbp = 0;

				// ---- Intra 4x4, lane = sample. The 16 blocks of a macroblock depend on their left, upper
				// and (where coded earlier) upper-right blocks: a chain of 10, not 16, when blocks
				// {2,4} {3,5} {6,8} {7,9} {10,12} {11,13} run side by side. This wave takes the first
				// of each pair, the chroma wave of the triple the second; they meet through s_i4a / s_i4b.
				// (Measured: the Intra4x4 part of a step 20500 -> 14900 cycles. Giving the chroma wave
				// the Intra16x16 macroblocks as well makes IT the long pole: 2.67 -> 2.80 ms.)
				if (__any(is4)) {
					if (is4)
						cf[COEF_DC16 + l] = 0;
					const int key = (base + t) * 16;
					const I4Ctx cx{ L.t, SY, cf_lane, nnz_mb, modes64, top_mb, left_mb, tr_mb, is4, g, l, pi, pj, zz,
							my_mf, my_dv, f_intra, q_qbits, q_qshift };
					uint32_t tab = i4_tab(s_ptab, cx, 0);
					cbp |= i4_block(s_ptab, cx, ci, cj, 0, 1, tab);
					cbp |= i4_block(s_ptab, cx, ci, cj, 1, 2, tab);
					lds_signal(&s_i4a[trip], key + 2, lane);
					cbp |= i4_block(s_ptab, cx, ci, cj, 2, 3, tab);
					lds_signal(&s_i4a[trip], key + 3, lane);
					cbp |= i4_block(s_ptab, cx, ci, cj, 3, 6, tab);
					lds_signal(&s_i4a[trip], key + 4, lane);
					lds_wait_ge(&s_i4b[trip], key + 4, xw); // blocks 4, 5
					cbp |= i4_block(s_ptab, cx, ci, cj, 6, 7, tab);
					lds_signal(&s_i4a[trip], key + 5, lane);
					cbp |= i4_block(s_ptab, cx, ci, cj, 7, 10, tab);
					lds_signal(&s_i4a[trip], key + 6, lane);
					lds_wait_ge(&s_i4b[trip], key + 6, xw); // blocks 8, 9
					cbp |= i4_block(s_ptab, cx, ci, cj, 10, 11, tab);
					cbp |= i4_block(s_ptab, cx, ci, cj, 11, 14, tab);
					lds_wait_ge(&s_i4b[trip], key + 8, xw); // blocks 12, 13
					cbp |= i4_block(s_ptab, cx, ci, cj, 14, 15, tab);
					cbp |= i4_block(s_ptab, cx, ci, cj, 15, 0, tab);
				}

				// ---- Intra 16x16: lane = 4x4 block (coding order) ----
				if (__any(is16)) {
					const int bx = blk_x(l), by = blk_y(l);
					const uint8_t *T = L.t;
					int d[16];
					int nn = 0, dcv = 128, pa = 0, pb = 0, pc = 0;
					auto pred16 = [&](int x, int y) {
						return i16_mode == 0 ? (int)T[8 + x]
						     : i16_mode == 1 ? (int)T[(1 + y) * IT_STRIDE + 7]
						     : i16_mode == 2 ? dcv
						     : clip255((pa + pb * (x - 7) + pc * (y - 7) + 16) >> 5);
					};
					if (is16) {
						int dcsum = 0, H = 0, V = 0;
						for (int k = 0; k < 16; k++)
							dcsum += (top_mb ? T[8 + k] : 0) + (left_mb ? T[(1 + k) * IT_STRIDE + 7] : 0);
						dcv = (top_mb && left_mb) ? (dcsum + 16) >> 5 : (top_mb || left_mb) ? (dcsum + 8) >> 4 : 128;
						for (int k = 0; k < 8; k++) {
							H += (k + 1) * ((int)T[8 + 8 + k] - (int)T[8 + 6 - k]);
							V += (k + 1) * ((int)T[(1 + 8 + k) * IT_STRIDE + 7] - (int)T[(1 + 6 - k) * IT_STRIDE + 7]);
						}
						pa = 16 * ((int)T[16 * IT_STRIDE + 7] + (int)T[8 + 15]);
						pb = (5 * H + 32) >> 6;
						pc = (5 * V + 32) >> 6;
						int w[16];
						{
							int res[16];
#pragma unroll
							for (int i = 0; i < 16; i++) {
								const int x = 4 * bx + (i & 3), y = 4 * by + (i >> 2);
								res[i] = (int)SY[y * 16 + x] - pred16(x, y);
							}
							fdct4x4(res, w);
						}
						L.dc[bx + 4 * by] = w[0];
						cf[COEF_LUMA + l * 16] = 0;
#pragma unroll
						for (int i = 1; i < 16; i++) {
							const int r = kZigzag[i], c = kPosClass[r];
							const int lq = quant1(w[r], q_mf[c], f_intra, q_qbits);
							cf[COEF_LUMA + l * 16 + i] = (int16_t)lq;
							nn += lq != 0;
							d[r] = (lq * q_dv[c]) << q_qshift;
						}
					}
					WAVE_SYNC();
					if (is16) { // forward Hadamard output (i,j) = raster lane
						const int i = l >> 2, j = l & 3;
						int s = 0;
#pragma unroll
						for (int a = 0; a < 4; a++)
#pragma unroll
							for (int b = 0; b < 4; b++)
								s += hsign(i, a) * hsign(j, b) * L.dc[4 * a + b];
						L.dcl[l] = quant1((s + 1) >> 1, q_mf[0], 2 * f_intra, q_qbits + 1);
					}
					WAVE_SYNC();
					if (is16) {
						cf[COEF_DC16 + l] = (int16_t)L.dcl[kZigzag[l]];
						int s = 0;
#pragma unroll
						for (int a = 0; a < 4; a++)
#pragma unroll
							for (int b = 0; b < 4; b++)
								s += hsign(by, a) * hsign(bx, b) * L.dcl[4 * a + b];
						const int ls = 16 * q_dv[0];
						d[0] = q_qshift >= 6 ? (s * ls) << (q_qshift - 6) : (s * ls + (1 << (5 - q_qshift))) >> (6 - q_qshift);
						int pred[16], out[16];
#pragma unroll
						for (int i = 0; i < 16; i++)
							pred[i] = pred16(4 * bx + (i & 3), 4 * by + (i >> 2));
						idct4x4_add(d, pred, out);
						// all predictions of the group are read before any lane writes the tile
						__builtin_amdgcn_wave_barrier();
#pragma unroll
						for (int k = 0; k < 16; k++)
							L.t[(1 + 4 * by + (k >> 2)) * IT_STRIDE + 8 + 4 * bx + (k & 3)] = (uint8_t)out[k];
						nnz_mb[bx + 4 * by] = (uint8_t)nn;
					}
					const uint64_t bal = __ballot(is16 && nn != 0);
					if (is16)
						cbp = ((uint32_t)(bal >> (16 * g)) & 0xffffu) ? 15 : 0;
				}
				// ---- I_PCM (k_intra_decide's open-loop count said the macroblock could pass 3200 bits): its
				// reconstruction is its source, what its neighbours predict from; no level, and every block counts
				// 16 coefficients for the neighbours' nC (9.2.1). lane = sample row / block ----
				const bool ispcm = act && mb_type == MB_IPCM;
				if (__any(ispcm)) {
					if (ispcm) {
						*(u32x4 *)(L.t + (1 + l) * IT_STRIDE + 8) = *(const u32x4 *)(SY + 16 * l);
						*(u32x4 *)(cf + COEF_LUMA + 16 * l) = u32x4{ 0, 0, 0, 0 };
						*(u32x4 *)(cf + COEF_LUMA + 16 * l + 8) = u32x4{ 0, 0, 0, 0 };
						cf[COEF_DC16 + l] = 0;
						nnz_mb[l] = 16;
					}
				}
				WAVE_SYNC();
				if (act) { // hand the macroblock to the mover and its last row to the group below
					*(u32x4 *)(L.outy[slot] + l * 16) = *(const u32x4 *)(L.t + (1 + l) * IT_STRIDE + 8);
					if (l == 15) {
						const u32x4 last = *(const u32x4 *)(L.t + 16 * IT_STRIDE + 8);
						*(u32x4 *)L.boty[mx & 3] = last;
						if (g == 3 && in_bot)
							*(u32x4 *)(rowy_mine + 16 + mx * 16) = last;
					}
					if (l == 0)
						L.cbp_y[slot] = cbp;
				}
				lds_signal(&s_done_y[trip], base + t + 1, lane);
			}
			continue;
		}

		if (role == 1) {
			// =========================== chroma wave ===========================
			for (int t = 0; t < nsteps; t++) {
				const int mx = t - 2 * g;
				const bool act = row_ok && mx >= 0 && mx < p.mb_w;
				const bool left_mb = mx > 0;
				const int slot = t % 3;
				const size_t mbaddr = (size_t)st * p.n_mb + (size_t)(act ? my : r0) * p.mb_w + (act ? mx : 0);
				lds_wait_ge(&s_staged[trip], base + t + 1, xw);
				load_q(slot);
				const int chroma_mode = (int)((L.info[slot][0] >> 16) & 0xff);
				const bool ispcm = (int)(int8_t)(L.info[slot][0] & 0xff) == MB_IPCM; // (see the luma wave)
				const uint8_t *SC = L.srcc[slot];
				if (in_top && t < p.mb_w) // only the macroblock above is needed here, but the luma wave's bound is as good
					lds_wait_ge(&s_done_c[trip_up], base_up + (t + 7 < nsteps ? t + 7 : nsteps), xw);
				if (act) {
					if (left_mb && l < 8)
						*(uint16_t *)(L.c + (1 + l) * IC_STRIDE + 6) = *(const uint16_t *)(L.c + (1 + l) * IC_STRIDE + 22);
					if (top_mb) {
						const uint8_t *rp = rowc_up + 16 + mx * 16;
						const uint8_t *tp = g > 0 ? LU.botc[mx & 3] : in_top ? rp : L.topc[slot] + 8;
						if (l == 8)
							*(u32x4 *)(L.c + 8) = *(const u32x4 *)tp;
						else if (l == 9 && left_mb)
							*(uint16_t *)(L.c + 6) = g > 0 ? *(const uint16_t *)(LU.botc[(mx - 1) & 3] + 14)
									       : in_top ? *(const uint16_t *)(rp - 2) : *(const uint16_t *)(L.topc[slot] + 6);
					}
				}
				WAVE_SYNC();
				{
					// the second block of each Intra4x4 pair (see the luma wave), on the luma tile
					const uint32_t miw = L.info[slot][0];
					const bool is4 = act && (int)(int8_t)(miw & 0xff) == MB_I4x4;
					int cbp_b = 0;
					if (__any(is4)) {
						const uint64_t modes64 = (uint64_t)L.info[slot][2] | (uint64_t)L.info[slot][3] << 32;
						const bool tr_mb = top_mb && mx < p.mb_w - 1;
						const int key = (base + t) * 16;
						const I4Ctx cx{ L.t, L.srcy[slot], p.coef + mbaddr * COEF_PER_MB + COEF_LUMA + zz, p.nnz + mbaddr * NNZ_PER_MB,
								modes64, top_mb, left_mb, tr_mb, is4, g, l, pi, pj, zz,
								my_mf, my_dv, f_intra, q_qbits, q_qshift };
						uint32_t tab = i4_tab(s_ptab, cx, 4);
						lds_wait_ge(&s_i4a[trip], key + 2, xw); // tile neighbours set up, blocks 0, 1
						cbp_b |= i4_block(s_ptab, cx, ci, cj, 4, 5, tab);
						cbp_b |= i4_block(s_ptab, cx, ci, cj, 5, 8, tab);
						lds_signal(&s_i4b[trip], key + 4, lane);
						lds_wait_ge(&s_i4a[trip], key + 4, xw); // blocks 2, 3
						cbp_b |= i4_block(s_ptab, cx, ci, cj, 8, 9, tab);
						lds_wait_ge(&s_i4a[trip], key + 5, xw); // block 6
						cbp_b |= i4_block(s_ptab, cx, ci, cj, 9, 12, tab);
						lds_signal(&s_i4b[trip], key + 6, lane);
						lds_wait_ge(&s_i4a[trip], key + 6, xw); // block 7
						cbp_b |= i4_block(s_ptab, cx, ci, cj, 12, 13, tab);
						cbp_b |= i4_block(s_ptab, cx, ci, cj, 13, 4, tab);
						lds_signal(&s_i4b[trip], key + 8, lane);
					}
					if (act && l == 10)
						L.cbp_yb[slot] = cbp_b;
				}
				// lane = sample of a 4x4 block (as the Intra4x4 path): the group walks the 8 chroma
				// blocks of its macroblock twice, transforms across lanes with DPP. Prediction only
				// reads the row above and the column left of the macroblock, so nothing written here
				// is read back within the step.
				int16_t *cf = p.coef + mbaddr * COEF_PER_MB;
				const uint8_t *T = L.c;
				int plane_a[2] = { 0, 0 }, plane_b[2] = { 0, 0 }, plane_c[2] = { 0, 0 };
				if (__any(act && chroma_mode == 3)) {
#pragma unroll
					for (int c = 0; c < 2; c++) {
						int H = 0, V = 0;
#pragma unroll
						for (int k = 0; k < 4; k++) {
							H += (k + 1) * ((int)T[8 + 2 * (4 + k) + c] - (int)T[8 + 2 * (2 - k) + c]);
							V += (k + 1) * ((int)T[(1 + 4 + k) * IC_STRIDE + 6 + c] - (int)T[(1 + 2 - k) * IC_STRIDE + 6 + c]);
						}
						plane_a[c] = 16 * ((int)T[(1 + 7) * IC_STRIDE + 6 + c] + (int)T[8 + 2 * 7 + c]);
						plane_b[c] = (34 * H + 32) >> 6;
						plane_c[c] = (34 * V + 32) >> 6;
					}
				}
				int pvs[8], cdq[8];
				uint32_t ac_any = 0;
#pragma unroll
				for (int it = 0; it < 8; it++) {
					const int c = it >> 2, blk = it & 3, bx = blk & 1, by = blk >> 1;
					const int x = 4 * bx + pj, y = 4 * by + pi;
					// DC of this block (clause 8.3.4.1-3): 4 samples above / left of the block
					const uint32_t m = 0x00ff00ffu << (8 * c);
					const uint32_t t0 = *(const uint32_t *)(T + 8 + 8 * bx), t1 = *(const uint32_t *)(T + 12 + 8 * bx);
					const int stp = (int)__builtin_amdgcn_sad_u8(t0 & m, 0u, 0u) + (int)__builtin_amdgcn_sad_u8(t1 & m, 0u, 0u);
					const int slf = (int)T[(1 + 4 * by) * IC_STRIDE + 6 + c] + (int)T[(2 + 4 * by) * IC_STRIDE + 6 + c] +
							(int)T[(3 + 4 * by) * IC_STRIDE + 6 + c] + (int)T[(4 + 4 * by) * IC_STRIDE + 6 + c];
					bool use_t = top_mb, use_l = left_mb;
					if (bx == 1 && by == 0 && top_mb)
						use_l = false;
					if (bx == 0 && by == 1 && left_mb)
						use_t = false;
					const int dcv = (use_t && use_l) ? (stp + slf + 4) >> 3 : use_t ? (stp + 2) >> 2 : use_l ? (slf + 2) >> 2 : 128;
					const int ph = T[(1 + y) * IC_STRIDE + 6 + c], pvv = T[8 + 2 * x + c];
					const int pp = clip255((plane_a[c] + plane_b[c] * (x - 3) + plane_c[c] * (y - 3) + 16) >> 5);
					const int pv = chroma_mode == 0 ? dcv : chroma_mode == 1 ? ph : chroma_mode == 2 ? pvv : pp;
					const int res = (int)SC[y * 16 + 2 * x + c] - pv;
					const int w = fdct_lanes(res, ci, cj);
					if (act && l == 0)
						L.cdc[4 * c + blk] = w; // lane (0,0) holds the DC coefficient
					const int level = (l == 0 || ispcm) ? 0 : quant1(w, my_mfc, f_intra_c, q_qbits_c);
					if (act)
						cf[COEF_CAC + (4 * c + blk) * 16 + zz] = (int16_t)level;
					cdq[it] = (level * my_dvc) << q_qshift_c;
					pvs[it] = pv;
					const uint64_t bal = __ballot(act && level != 0);
					const int nn = __popc((uint32_t)(bal >> (16 * g)) & 0xffffu);
					if (act && l == 0)
						p.nnz[mbaddr * NNZ_PER_MB + 16 + 4 * c + blk] = (uint8_t)(ispcm ? 16 : nn);
					ac_any |= (uint32_t)nn;
				}
				WAVE_SYNC();
				int cdcnz = 0;
				if (act && l < 8) { // 2x2 Hadamard of the chroma DCs: lane = (plane, output index)
					const int c = l >> 2, blk = l & 3;
					const int *dcs = L.cdc + 4 * c;
					const int s1 = (blk & 1) ? -1 : 1, s2 = (blk & 2) ? -1 : 1;
					const int hd = dcs[0] + s1 * dcs[1] + s2 * dcs[2] + s1 * s2 * dcs[3];
					const int lq = ispcm ? 0 : quant1(hd, q_mfc0, 2 * f_intra_c, q_qbits_c + 1);
					L.cdl[l] = lq;
					cf[COEF_CDC + l] = (int16_t)lq;
					cdcnz = lq != 0;
				}
				WAVE_SYNC();
#pragma unroll
				for (int it = 0; it < 8; it++) {
					const int c = it >> 2, blk = it & 3, bx = blk & 1, by = blk >> 1;
					const int x = 4 * bx + pj, y = 4 * by + pi;
					const int *ls = L.cdl + 4 * c;
					const int s1 = (blk & 1) ? -1 : 1, s2 = (blk & 2) ? -1 : 1;
					const int fd = ls[0] + s1 * ls[1] + s2 * ls[2] + s1 * s2 * ls[3];
					const int dcr = ((fd * q_dvc0 * 16) << q_qshift_c) >> 5;
					const int r = idct_lanes(l == 0 ? dcr : cdq[it], ci, cj);
					if (act) // (an I_PCM macroblock: its source sample)
						L.c[(1 + y) * IC_STRIDE + 8 + 2 * x + c] = ispcm ? SC[y * 16 + 2 * x + c] : (uint8_t)clip255(pvs[it] + r);
				}
				const uint64_t dcm = __ballot(cdcnz != 0);
				const uint32_t ac = ac_any, dcb = (uint32_t)(dcm >> (16 * g)) & 0xffffu;
				WAVE_SYNC();
				if (act) {
					if (l < 8)
						*(u32x4 *)(L.outc[slot] + l * 16) = *(const u32x4 *)(L.c + (1 + l) * IC_STRIDE + 8);
					else if (l == 8) {
						const u32x4 last = *(const u32x4 *)(L.c + 8 * IC_STRIDE + 8);
						*(u32x4 *)L.botc[mx & 3] = last;
						if (g == 3 && in_bot)
							*(u32x4 *)(rowc_mine + 16 + mx * 16) = last;
					}
					else if (l == 9)
						L.cbp_c[slot] = ac ? 0x20 : (dcb ? 0x10 : 0);
				}
				lds_signal(&s_done_c[trip], base + t + 1, lane);
			}
			continue;
		}

		// ============================== mover wave ==============================
		struct MoverRegs {
			u32x4 sy, sc, modes;
			uint64_t mi, ta, tb;
			int qoff; // regions of interest / MAD classes: the macroblock's QP (k_intra_decide)
			bool top_ok;
		};
		MoverRegs RA{}, RB{};
		int pending_x = -1;

		auto load_step = [&](int t, bool block, MoverRegs &R) {
			const int nmx = t - 2 * g, mx0 = t;
			const bool nact = row_ok && nmx >= 0 && nmx < p.mb_w;
			const int cmx = nact ? nmx : 0, cmy = row_ok ? my : r0;
			const size_t a = (size_t)st * p.n_mb + (size_t)cmy * p.mb_w + cmx;
			const int x0 = cmx * 16, y0 = cmy * 16;
			const bool left_mb = nmx > 0, tr_mb = top_mb && nmx < p.mb_w - 1;
			R.top_ok = true;
			const int need = mx0 + 2 < p.mb_w ? mx0 + 2 : p.mb_w; // the macroblock above-right must be done
			if (xwg_band && mx0 < p.mb_w) { // rows above live in another workgroup

				if (xseen < (uint32_t)need)
					xseen = xload32(&xprog[wj - 1]);
				if (xseen < (uint32_t)need && block)
					xseen = xwait(&xprog[wj - 1], xabort, xerr, (uint32_t)need);
				R.top_ok = xseen >= (uint32_t)need;
			}
			R.sy = *(const u32x4 *)(src_y + (size_t)min(y0 + l, p.src_last_y) * p.cw + x0);
			R.sc = *(const u32x4 *)(src_uv + (size_t)min((y0 >> 1) + (l & 7), p.src_last_c) * p.cw + x0);
			R.mi = *(const uint64_t *)&p.mbi[a];
			R.modes = *(const u32x4 *)(p.i4modes + a * 16);
			if (ROI)
				R.qoff = (int)p.qp_t[a];
			R.ta = R.tb = 0;
			if (g == 0 && top_mb && nact && R.top_ok && !in_top) {
				if (xwg_top) { // the other workgroup's write-through buffer
					const uint64_t *hb = hb_in + (size_t)cmx * WF_HBUF_GRANULES;
					if (l == 8 || l == 9)
						R.ta = xload64(hb + (l - 8));
					else if (l == 10 && tr_mb)
						R.ta = xload64(hb + WF_HBUF_GRANULES);         // next MB, first 4 samples
					else if (l == 11 && left_mb)
						R.ta = xload64(hb - WF_HBUF_GRANULES + 1) >> 56; // previous MB, last sample
					else if (l == 12 || l == 13)
						R.ta = xload64(hb + 2 + (l - 12));
					else if (l == 14 && left_mb)
						R.ta = xload64(hb - WF_HBUF_GRANULES + 3) >> 48; // previous MB, last CbCr pair
				} else {
					if (l == 8) {
						const uint64_t *tp = (const uint64_t *)(rec_y + (size_t)(y0 - 1) * p.cw + x0);
						R.ta = tp[0];
						R.tb = tp[1];
					} else if (l == 10 && tr_mb) {
						R.ta = *(const uint32_t *)(rec_y + (size_t)(y0 - 1) * p.cw + x0 + 16);
					} else if (l == 11 && left_mb) {
						R.ta = rec_y[(size_t)(y0 - 1) * p.cw + x0 - 1];
					} else if (l == 12) {
						const uint64_t *tp = (const uint64_t *)(rec_uv + (size_t)((y0 >> 1) - 1) * p.cw + x0);
						R.ta = tp[0];
						R.tb = tp[1];
					} else if (l == 14 && left_mb) {
						R.ta = *(const uint16_t *)(rec_uv + (size_t)((y0 >> 1) - 1) * p.cw + x0 - 2);
					}
				}
			}
		};

		auto stage_step = [&](int t, const MoverRegs &R) {
			const int mx = t - 2 * g;
			const bool act = row_ok && mx >= 0 && mx < p.mb_w;
			const int slot = t % 3;
			if (!act)
				return;
			*(u32x4 *)(L.srcy[slot] + l * 16) = R.sy;
			if (l < 8)
				*(u32x4 *)(L.srcc[slot] + l * 16) = R.sc;
			if (l == 15) {
				const uint32_t mw[4] = { R.modes.x, R.modes.y, R.modes.z, R.modes.w };
				uint64_t modes64 = 0;
#pragma unroll
				for (int k = 0; k < 16; k++)
					modes64 |= (uint64_t)((mw[k >> 2] >> (8 * (k & 3))) & 15u) << (4 * k);
				*(u32x4 *)L.info[slot] = u32x4{ (uint32_t)R.mi, (uint32_t)(R.mi >> 32), (uint32_t)modes64, (uint32_t)(modes64 >> 32) };
				if (ROI)
					L.qpw[slot] = (uint32_t)R.qoff;
			}
			if (g == 0 && top_mb && !in_top) {
				if (xwg_top) {
					if (l == 8 || l == 9)
						*(uint64_t *)(L.topy[slot] + 8 + (l - 8) * 8) = R.ta;
					else if (l == 10)
						*(uint32_t *)(L.topy[slot] + 24) = (uint32_t)R.ta;
					else if (l == 11)
						L.topy[slot][7] = (uint8_t)R.ta;
					else if (l == 12 || l == 13)
						*(uint64_t *)(L.topc[slot] + 8 + (l - 12) * 8) = R.ta;
					else if (l == 14)
						*(uint16_t *)(L.topc[slot] + 6) = (uint16_t)R.ta;
				} else {
					if (l == 8) {
						*(uint64_t *)(L.topy[slot] + 8) = R.ta;
						*(uint64_t *)(L.topy[slot] + 16) = R.tb;
					} else if (l == 10) {
						*(uint32_t *)(L.topy[slot] + 24) = (uint32_t)R.ta;
					} else if (l == 11) {
						L.topy[slot][7] = (uint8_t)R.ta;
					} else if (l == 12) {
						*(uint64_t *)(L.topc[slot] + 8) = R.ta;
						*(uint64_t *)(L.topc[slot] + 16) = R.tb;
					} else if (l == 14) {
						*(uint16_t *)(L.topc[slot] + 6) = (uint16_t)R.ta;
					}
				}
			}
		};

		auto writeback_step = [&](int t) {
			const int mx = t - 2 * g;
			const bool act = row_ok && mx >= 0 && mx < p.mb_w;
			const int slot = t % 3;
			if (!act)
				return;
			const int x0 = mx * 16, y0 = my * 16;
			const size_t mbaddr = (size_t)st * p.n_mb + (size_t)my * p.mb_w + mx;
			*(u32x4 *)(rec_y + (size_t)(y0 + l) * p.cw + x0) = *(const u32x4 *)(L.outy[slot] + l * 16);
			if (l < 8)
				*(u32x4 *)(rec_uv + (size_t)(my * 8 + l) * p.cw + x0) = *(const u32x4 *)(L.outc[slot] + l * 16);
			if (l == 15) { // the record, with the coded block pattern of both compute waves
				uint32_t cbp = (uint32_t)(L.cbp_y[slot] | L.cbp_yb[slot] | L.cbp_c[slot]) & 0xffu;
				// macroblocks differ in QP: the first one of a row always carries an mb_qp_delta (an Intra4x4
				// one without a coded block says "chroma DC" and codes two empty blocks), so that QP_Y never
				// has to be inherited across a row boundary
				if (ROI && p.row_rule && mx == 0 && (int)(int8_t)(L.info[slot][0] & 0xffu) == MB_I4x4 && !(cbp & 0x3fu))
					cbp = 0x10u;
				const uint32_t w0 = (L.info[slot][0] & 0x00ffffffu) | cbp << 24;
				*(uint64_t *)&p.mbi[mbaddr] = (uint64_t)w0 | (uint64_t)L.info[slot][1] << 32;
			}
			if (is_xprod) {
				if (l < 4) {
					const uint8_t *srcp = l < 2 ? L.outy[slot] + 15 * 16 + l * 8 : L.outc[slot] + 7 * 16 + (l - 2) * 8;
					xstore64(hb_out + (size_t)mx * WF_HBUF_GRANULES + l, *(const uint64_t *)srcp);
				}
				pending_x = mx + 1;
			}
		};

		auto publish = [&]() { // everything issued an iteration ago has completed by now
			if (b_hi < nbands && band == b_hi - 1) {
				asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
				if (is_xprod && l == 0 && pending_x >= 0)
					xstore32(&xprog[wj], (uint32_t)pending_x);
				pending_x = -1;
			}
		};

		// frame pipelining: this picture's deblocking (another kernel, already resident) follows the
		// write-back of every band along the diagonals x + 2*y, see k_gate / deblock2_body
		const int diag0 = 2 * (r0 + 4 * band);
		auto publish_diag = [&](uint32_t d) {
			if (!p.prog_intra)
				return;
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
			if (lane == 0)
				__hip_atomic_store(p.prog_intra + unit * PROG_BANDS + (band < PROG_BANDS ? band : PROG_BANDS - 1),
						   p.prog_tag | d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		};
		auto mover_iter = [&](int t, MoverRegs &R) { // R holds step t+1 and is refilled with step t+3
			if (t >= 1) {
				lds_wait_ge(&s_done_y[trip], base + t, xw);
				lds_wait_ge(&s_done_c[trip], base + t, xw);
			}
			publish();
			if ((t & 7) == 0 && t >= 8) // steps <= t-2 were written back an iteration ago
				publish_diag((uint32_t)(t - 1 + diag0));
			if (t + 1 < nsteps) {
				if (!R.top_ok)
					load_step(t + 1, true, R);
				stage_step(t + 1, R);
				load_step(t + 3, false, R);
				const bool lost = p.fault_inject == 2 && unit == 0 && band == 0 && t >= 8; // fault injection (tests)
				lds_signal(&s_staged[trip], lost ? base + 1 : base + t + 2, lane);
			}
			if (t >= 1)
				writeback_step(t - 1);
		};
		load_step(0, true, RA);
		load_step(1, false, RB);
		stage_step(0, RA);
		load_step(2, false, RA);
		lds_signal(&s_staged[trip], base + 1, lane);
		for (int t = 0; t <= nsteps; t += 2) {
			mover_iter(t, RB);
			if (t + 1 <= nsteps)
				mover_iter(t + 1, RA);
		}
		publish();
		publish_diag((uint32_t)PROG_ALL);
	}
}


template <int PAIRS>
static int launch_deblock(const Params &p, hipStream_t s, int planes)
{
	const int nwg = p.n_streams * p.slice_num * p.wf_split;
	// dynamic LDS of what the workgroups of this launch filter (planes 4: luma workgroups and chroma workgroups,
	// the bigger of the two). 8 pairs, both planes: ~134 KiB, one workgroup per CU by construction; 2 pairs:
	// 37 KiB for both planes in one workgroup, 25 KiB for luma, 13 KiB for chroma
	constexpr size_t l1 = sizeof(Db2Shared<1, PAIRS>), l2 = sizeof(Db2Shared<2, PAIRS>), l3 = sizeof(Db2Shared<3, PAIRS>);
	const size_t lds = planes == 3 ? l3 : planes == 1 ? l1 : planes == 2 ? l2 : (l1 > l2 ? l1 : l2);
	const size_t lds_max = l3 > l1 ? l3 : l1;
	static_assert(l1 <= 160 * 1024 && l3 <= 160 * 1024 && l2 <= l3, "a CU has 160 KiB of LDS");
	static DeviceOnce attr; // (one per instantiation of this launcher)
	auto dyn = [&](const void *k) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max) == hipSuccess; };
	if (!attr.run([&]() {
		    return dyn((const void *)k_deblock2_planes<0, PAIRS>) && dyn((const void *)k_deblock2_planes<1, PAIRS>) && dyn((const void *)k_deblock2_planes<2, PAIRS>) &&
			   dyn((const void *)k_deblock2_planes<3, PAIRS>) && dyn((const void *)k_deblock2_planes<4, PAIRS>) && dyn((const void *)k_deblock2_planes<5, PAIRS>);
	    }))
		return -1;
	dim3 grid(planes == 4 ? 2 * nwg : nwg);
	if (p.roi_off) // regions of interest: thresholds per macroblock; an IDR picture is "every macroblock intra" there
		hipLaunchKernelGGL((k_deblock2_planes<3, PAIRS>), grid, dim3(128 * PAIRS), lds, s, p, planes);
	else if (p.pcm && p.is_idr) // I_PCM macroblocks may be among them: their edges are left alone
		hipLaunchKernelGGL((k_deblock2_planes<5, PAIRS>), grid, dim3(128 * PAIRS), lds, s, p, planes);
	else if (p.pcm)
		hipLaunchKernelGGL((k_deblock2_planes<4, PAIRS>), grid, dim3(128 * PAIRS), lds, s, p, planes);
	else if (p.is_idr)
		hipLaunchKernelGGL((k_deblock2_planes<1, PAIRS>), grid, dim3(128 * PAIRS), lds, s, p, planes);
	else if (p.intra_refresh > 0) // intra macroblocks among the inter ones: both sets of boundary strengths
		hipLaunchKernelGGL((k_deblock2_planes<2, PAIRS>), grid, dim3(128 * PAIRS), lds, s, p, planes);
	else
		hipLaunchKernelGGL((k_deblock2_planes<0, PAIRS>), grid, dim3(128 * PAIRS), lds, s, p, planes);
	return (int)hipGetLastError();
}

int Launch::deblock(const Params &p, void *stream, int planes)
{
	if (p.db_rows)
		return launch_deblock_rows(p, stream);
	return p.db_pairs == 2 ? launch_deblock<2>(p, (hipStream_t)stream, planes) : launch_deblock<DB2_PAIRS_MAX>(p, (hipStream_t)stream, planes);
}

int Launch::intra_recon(const Params &p, void *stream)
{
	if (p.ir_rows)
		return launch_intra_rows(p, stream);
	dim3 grid(p.n_streams * p.slice_num * p.wf_split_intra);
	// wavefront workgroups are throughput-bound on their CU: enough (unused dynamic) LDS that
	// two of them can never share one CU (160 KiB per CU)
	// dynamic LDS: the band-to-band row buffers, padded so that two workgroups never share a CU
	const size_t rows = (size_t)2 * IR_TRIPLES * (p.mb_w * 16 + 32);
	if (p.roi_off)
		hipLaunchKernelGGL(k_intra_recon2<true>, grid, dim3(64 * 3 * IR_TRIPLES), rows > 64 * 1024 ? rows : 64 * 1024, (hipStream_t)stream, p);
	else
		hipLaunchKernelGGL(k_intra_recon2<false>, grid, dim3(64 * 3 * IR_TRIPLES), rows > 64 * 1024 ? rows : 64 * 1024, (hipStream_t)stream, p);
	return (int)hipGetLastError();
}

} // namespace cedar
