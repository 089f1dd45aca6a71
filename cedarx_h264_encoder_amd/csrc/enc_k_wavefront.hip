// cedarx_h264_encoder_amd/csrc/enc_k_wavefront.hip
// The two macroblock-ORDERED stages of H.264 as gfx950 wavefront kernels:
//   k_intra_recon  closed-loop intra prediction + transform + reconstruction
//                  (needs left / top / top-right reconstructed neighbours)
//   k_deblock      in-loop deblocking filter, clause 8.7, raster MB order
//                  semantics (an MB's filters read samples its left / top /
//                  top-right neighbours have already filtered)
// Both have the dependency (x, y) <- (x-1, y), (x, y-1), (x+1, y-1), i.e. a 2-MB
// lag per row. One workgroup of WF_WAVES waves owns one unit (stream, slice);
// rows are dealt to waves and every row publishes its progress in LDS, so a
// row proceeds as soon as the row above is two macroblocks ahead. All
// synchronisation is workgroup scope (one CU): no cross-CU hand-off, no grid
// barrier, nothing that can deadlock on residency. Many units = many CUs.
//
// The reference delegates both stages to silicon (kernel/cedar.c:1135-1153
// programs REC/REF/MBINFO addresses, :1176 triggers); deblocking policy is the
// slice header's (cedar.c:1024-1029: idc 0, offsets 0).
#include <hip/hip_runtime.h>

#include "enc_device.h"

namespace cedar {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define WAVE_SYNC()                                                   \
	do {                                                          \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); \
		__builtin_amdgcn_wave_barrier();                      \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); \
	} while (0)

constexpr int MAX_ROWS = 512; // MB rows of one unit (8K pictures)

DEVI void wait_progress(volatile int *prog, int idx, int need)
{
	// wave-uniform spin on an LDS word written by another wave of this workgroup
	while (__hip_atomic_load((int *)&prog[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < need)
		__builtin_amdgcn_s_sleep(1);
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

DEVI void publish_progress(volatile int *prog, int idx, int value)
{
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
	__hip_atomic_store((int *)&prog[idx], value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// 4x4 Hadamard sign H[i][a] for rows (1,1,1,1) (1,1,-1,-1) (1,-1,-1,1) (1,-1,1,-1)
DEVI int hsign(int i, int a)
{
	// bit a of nibble i set = minus. row0 ++++ (0000), row1 ++-- (1100),
	// row2 +--+ (0110), row3 +-+- (1010)
	const uint32_t neg = (0xAu << 12) | (0x6u << 8) | (0xCu << 4) | 0x0u;
	return ((neg >> (4 * i + a)) & 1) ? -1 : 1;
}

// ------------------------------------------------------------------------
// k_intra_recon
// ------------------------------------------------------------------------
#define IT_STRIDE 48 // luma tile: sample (x,y) at [(1+y)*48 + 16 + x], x in [-1,19], y in [-1,15]
#define IC_STRIDE 32 // chroma tile (nv12): sample (x,y,c) at [(1+y)*32 + 16 + 2x + c]

struct IntraWaveLds {
	uint8_t t[17 * IT_STRIDE];
	uint8_t c[9 * IC_STRIDE];
	uint8_t src[256];
	uint8_t srcc[128];
	uint8_t pred[256];
	uint8_t predc[2][64];
	int16_t coef[COEF_PER_MB];
	uint8_t E[16];
	int16_t res4[16];
	uint8_t pred4[16];
	int dc[16];
	int dcl[16];
	int cdc[8];
	int cdl[8];
	uint8_t nnz[24];
	uint8_t pad[8];
};

__global__ __launch_bounds__(64 * WF_WAVES) void k_intra_recon(Params p)
{
	__shared__ __attribute__((aligned(16))) IntraWaveLds s_w[WF_WAVES];
	__shared__ int s_prog[MAX_ROWS];

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int unit = blockIdx.x;
	const int st = unit / p.slice_num, sl = p.slice_first + unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, sl, &r0, &r1);
	for (int i = threadIdx.x; i < MAX_ROWS; i += blockDim.x)
		s_prog[i] = 0;
	__syncthreads();

	IntraWaveLds &L = s_w[wave];
	const uint8_t *src_y = p.src_y + (size_t)st * p.plane_y_stride;
	const uint8_t *src_uv = p.src_uv + (size_t)st * p.plane_uv_stride;
	uint8_t *rec_y = p.rec_y + (size_t)st * p.plane_y_stride;
	uint8_t *rec_uv = p.rec_uv + (size_t)st * p.plane_uv_stride;
	const int f_intra = (1 << p.qbits) / 3, f_intra_c = (1 << p.qbits_c) / 3;

	for (int my = r0 + wave; my < r1; my += WF_WAVES) {
		const bool top_mb = my > r0;
		for (int mx = 0; mx < p.mb_w; mx++) {
			const int x0 = mx * 16, y0 = my * 16;
			const bool left_mb = mx > 0, tr_mb = top_mb && mx < p.mb_w - 1;
			const size_t mbaddr = (size_t)st * p.n_mb + (size_t)my * p.mb_w + mx;

			// left neighbours = last column of the previous macroblock, still in the tiles
			if (left_mb) {
				if (lane < 16)
					L.t[(1 + lane) * IT_STRIDE + 15] = L.t[(1 + lane) * IT_STRIDE + 31];
				else if (lane < 24) {
					int y = lane - 16;
					L.c[(1 + y) * IC_STRIDE + 14] = L.c[(1 + y) * IC_STRIDE + 30];
					L.c[(1 + y) * IC_STRIDE + 15] = L.c[(1 + y) * IC_STRIDE + 31];
				}
			}
			if (top_mb)
				wait_progress(s_prog, my - 1 - r0, mx + 2 < p.mb_w ? mx + 2 : p.mb_w);
			// source macroblock + top neighbours from the (unfiltered) reconstruction
			if (lane < 16)
				*(u32x4 *)(L.src + lane * 16) = *(const u32x4 *)(src_y + (size_t)(y0 + lane) * p.cw + x0);
			else if (lane < 24)
				*(u32x4 *)(L.srcc + (lane - 16) * 16) = *(const u32x4 *)(src_uv + (size_t)(my * 8 + lane - 16) * p.cw + x0);
			else if (top_mb) {
				if (lane == 24)
					*(u32x4 *)(L.t + 16) = *(const u32x4 *)(rec_y + (size_t)(y0 - 1) * p.cw + x0);
				else if (lane == 25 && tr_mb)
					*(uint32_t *)(L.t + 32) = *(const uint32_t *)(rec_y + (size_t)(y0 - 1) * p.cw + x0 + 16);
				else if (lane == 26 && left_mb)
					L.t[15] = rec_y[(size_t)(y0 - 1) * p.cw + x0 - 1];
				else if (lane == 27)
					*(u32x4 *)(L.c + 16) = *(const u32x4 *)(rec_uv + (size_t)(my * 8 - 1) * p.cw + x0);
				else if (lane == 28 && left_mb)
					*(uint16_t *)(L.c + 14) = *(const uint16_t *)(rec_uv + (size_t)(my * 8 - 1) * p.cw + x0 - 2);
			}
			const MbInfo mi = p.mbi[mbaddr];
			const int my_i4 = lane < 16 ? p.i4modes[mbaddr * 16 + lane] : 0;
			for (int i = lane; i < COEF_PER_MB; i += 64)
				L.coef[i] = 0;
			if (lane < 24)
				L.nnz[lane] = 0;
			WAVE_SYNC();

			int cbp = 0;
			if (mi.type == MB_I16x16) {
				// ---- 16x16 prediction: lane covers row lane>>2, columns 4*(lane&3).. ----
				{
					const int py = lane >> 2, pxb = 4 * (lane & 3);
					int dcsum = 0, H = 0, V = 0;
					for (int k = 0; k < 16; k++)
						dcsum += (top_mb ? L.t[16 + k] : 0) + (left_mb ? L.t[(1 + k) * IT_STRIDE + 15] : 0);
					const int dcv = (top_mb && left_mb) ? (dcsum + 16) >> 5 : (top_mb || left_mb) ? (dcsum + 8) >> 4 : 128;
					for (int k = 0; k < 8; k++) {
						H += (k + 1) * ((int)L.t[16 + 8 + k] - (int)L.t[16 + 6 - k]);
						V += (k + 1) * ((int)L.t[(1 + 8 + k) * IT_STRIDE + 15] - (int)L.t[(1 + 6 - k) * IT_STRIDE + 15]);
					}
					const int a = 16 * ((int)L.t[16 * IT_STRIDE + 15] + (int)L.t[16 + 15]);
					const int b = (5 * H + 32) >> 6, cc = (5 * V + 32) >> 6;
#pragma unroll
					for (int k = 0; k < 4; k++) {
						const int px = pxb + k;
						int v;
						if (mi.i16_mode == 0)
							v = L.t[16 + px];
						else if (mi.i16_mode == 1)
							v = L.t[(1 + py) * IT_STRIDE + 15];
						else if (mi.i16_mode == 2)
							v = dcv;
						else
							v = clip255((a + b * (px - 7) + cc * (py - 7) + 16) >> 5);
						L.pred[py * 16 + px] = (uint8_t)v;
					}
				}
				WAVE_SYNC();
				int pred[16], w[16], lv[16], d[16], out[16];
				int nn = 0;
				const int bx = kBlkX[lane & 15], by = kBlkY[lane & 15];
				if (lane < 16) {
					int res[16];
#pragma unroll
					for (int i = 0; i < 16; i++) {
						int x = 4 * bx + (i & 3), y = 4 * by + (i >> 2);
						pred[i] = L.pred[y * 16 + x];
						res[i] = (int)L.src[y * 16 + x] - pred[i];
					}
					fdct4x4(res, w);
					L.dc[bx + 4 * by] = w[0];
					nn = quant_dequant(w, 1, p.mf, p.dv, f_intra, p.qbits, p.qshift, lv, d);
				}
				WAVE_SYNC();
				if (lane < 16) { // forward Hadamard output (i,j) = raster lane
					const int i = lane >> 2, j = lane & 3;
					int s = 0;
#pragma unroll
					for (int a = 0; a < 4; a++)
#pragma unroll
						for (int b = 0; b < 4; b++)
							s += hsign(i, a) * hsign(j, b) * L.dc[4 * a + b];
					L.dcl[lane] = quant1((s + 1) >> 1, p.mf[0], 2 * f_intra, p.qbits + 1);
				}
				WAVE_SYNC();
				if (lane < 16) {
					// zig-zag position `lane` of the DC block holds raster kZigzag[lane]
					L.coef[COEF_DC16 + lane] = (int16_t)L.dcl[kZigzag[lane]];
					// inverse Hadamard + scaling for this lane's own block position
					const int i = by, j = bx;
					int s = 0;
#pragma unroll
					for (int a = 0; a < 4; a++)
#pragma unroll
						for (int b = 0; b < 4; b++)
							s += hsign(i, a) * hsign(j, b) * L.dcl[4 * a + b];
					const int ls = 16 * p.dv[0];
					d[0] = p.qp >= 36 ? (s * ls) << (p.qshift - 6) : (s * ls + (1 << (5 - p.qshift))) >> (6 - p.qshift);
					idct4x4_add(d, pred, out);
#pragma unroll
					for (int k = 1; k < 16; k++)
						L.coef[COEF_LUMA + lane * 16 + k] = (int16_t)lv[k];
#pragma unroll
					for (int k = 0; k < 16; k++)
						L.t[(1 + 4 * by + (k >> 2)) * IT_STRIDE + 16 + 4 * bx + (k & 3)] = (uint8_t)out[k];
					L.nnz[bx + 4 * by] = (uint8_t)nn;
				}
				cbp = (__ballot(nn != 0) & 0xffffull) ? 15 : 0;
			} else {
				// ---- Intra 4x4: blocks strictly in coding order ----
				for (int idx = 0; idx < 16; idx++) {
					const int bx = kBlkX[idx], by = kBlkY[idx];
					const int tx = 16 + 4 * bx, ty = 1 + 4 * by;
					bool ht, hl, htr;
					blk4_avail(idx, top_mb, left_mb, tr_mb, ht, hl, htr);
					if (lane < 15) {
						// E[0]=E[1]=l3 E[2]=l2 E[3]=l1 E[4]=l0 E[5]=tl E[6..13]=t0..t7 E[14]=t7
						int v;
						if (lane <= 4) {
							int k = lane == 0 ? 3 : 4 - lane; // left row index
							v = L.t[(ty + k) * IT_STRIDE + tx - 1];
						} else if (lane == 5) {
							v = L.t[(ty - 1) * IT_STRIDE + tx - 1];
						} else {
							int k = lane - 6; // top index 0..8 (8 = duplicate of 7)
							k = k > 7 ? 7 : k;
							k = (k > 3 && !htr) ? 3 : k;
							v = L.t[(ty - 1) * IT_STRIDE + tx + k];
						}
						L.E[lane] = (uint8_t)v;
					}
					WAVE_SYNC();
					const int mode = __shfl(my_i4, idx, 64);
					if (lane < 16) {
						const int st_ = L.E[6] + L.E[7] + L.E[8] + L.E[9], sl_ = L.E[1] + L.E[2] + L.E[3] + L.E[4];
						const int dcv = (ht && hl) ? (st_ + sl_ + 4) >> 3 : hl ? (sl_ + 2) >> 2 : ht ? (st_ + 2) >> 2 : 128;
						const int pv = pred4_px(p.pred4_tab[mode * 16 + lane], L.E, dcv);
						L.pred4[lane] = (uint8_t)pv;
						L.res4[lane] = (int16_t)((int)L.src[(4 * by + (lane >> 2)) * 16 + 4 * bx + (lane & 3)] - pv);
					}
					WAVE_SYNC();
					if (lane == 0) {
						int pred[16], res[16], w[16], lv[16], d[16], out[16];
#pragma unroll
						for (int i = 0; i < 16; i++) {
							pred[i] = L.pred4[i];
							res[i] = L.res4[i];
						}
						fdct4x4(res, w);
						const int nn = quant_dequant(w, 0, p.mf, p.dv, f_intra, p.qbits, p.qshift, lv, d);
						idct4x4_add(d, pred, out);
#pragma unroll
						for (int i = 0; i < 16; i++) {
							L.coef[COEF_LUMA + idx * 16 + i] = (int16_t)lv[i];
							L.t[(ty + (i >> 2)) * IT_STRIDE + tx + (i & 3)] = (uint8_t)out[i];
						}
						L.nnz[bx + 4 * by] = (uint8_t)nn;
						cbp |= nn ? (1 << (idx >> 2)) : 0;
					}
					WAVE_SYNC();
				}
				cbp = __shfl(cbp, 0, 64);
			}

			// ---- chroma prediction: lane = sample (x = lane&7, y = lane>>3), both planes ----
			{
				const int px = lane & 7, py = lane >> 3;
#pragma unroll
				for (int c = 0; c < 2; c++) {
					const uint8_t *T = L.c;
#define CT(xx, yy) T[(1 + (yy)) * IC_STRIDE + 16 + 2 * (xx) + c]
					int v;
					if (mi.chroma_mode == 0) {
						const int bx = px >> 2, by = py >> 2;
						int stp = 0, slf = 0;
						for (int k = 0; k < 4; k++) {
							stp += CT(4 * bx + k, -1);
							slf += CT(-1, 4 * by + k);
						}
						bool use_t = top_mb, use_l = left_mb;
						if (bx == 1 && by == 0 && top_mb)
							use_l = false;
						if (bx == 0 && by == 1 && left_mb)
							use_t = false;
						v = (use_t && use_l) ? (stp + slf + 4) >> 3 : use_t ? (stp + 2) >> 2 : use_l ? (slf + 2) >> 2 : 128;
					} else if (mi.chroma_mode == 1) {
						v = CT(-1, py);
					} else if (mi.chroma_mode == 2) {
						v = CT(px, -1);
					} else {
						int H = 0, V = 0;
						for (int k = 0; k < 4; k++) {
							H += (k + 1) * ((int)CT(4 + k, -1) - (int)CT(2 - k, -1));
							V += (k + 1) * ((int)CT(-1, 4 + k) - (int)CT(-1, 2 - k));
						}
						const int a = 16 * ((int)CT(-1, 7) + (int)CT(7, -1));
						const int b = (34 * H + 32) >> 6, cc = (34 * V + 32) >> 6;
						v = clip255((a + b * (px - 3) + cc * (py - 3) + 16) >> 5);
					}
#undef CT
					L.predc[c][py * 8 + px] = (uint8_t)v;
				}
			}
			WAVE_SYNC();
			// ---- chroma residual: lanes 16..23 = (plane, block) ----
			int cnn = 0, cdcnz = 0;
			{
				const bool is_c = lane >= 16 && lane < 24;
				const int c = (lane - 16) >> 2, blk = (lane - 16) & 3;
				const int bx = blk & 1, by = blk >> 1;
				int cpred[16], cw_[16], clv[16], cd[16], out[16];
				if (is_c) {
					int res[16];
#pragma unroll
					for (int i = 0; i < 16; i++) {
						int x = 4 * bx + (i & 3), y = 4 * by + (i >> 2);
						cpred[i] = L.predc[c][y * 8 + x];
						res[i] = (int)L.srcc[y * 16 + 2 * x + c] - cpred[i];
					}
					fdct4x4(res, cw_);
					L.cdc[lane - 16] = cw_[0];
				}
				WAVE_SYNC();
				if (is_c) {
					const int *dcs = L.cdc + 4 * c;
					const int s1 = (blk & 1) ? -1 : 1, s2 = (blk & 2) ? -1 : 1;
					const int hd = dcs[0] + s1 * dcs[1] + s2 * dcs[2] + s1 * s2 * dcs[3];
					const int l = quant1(hd, p.mfc[0], 2 * f_intra_c, p.qbits_c + 1);
					L.cdl[lane - 16] = l;
					L.coef[COEF_CDC + 4 * c + blk] = (int16_t)l;
					cdcnz = l != 0;
				}
				WAVE_SYNC();
				if (is_c) {
					const int *ls = L.cdl + 4 * c;
					const int s1 = (blk & 1) ? -1 : 1, s2 = (blk & 2) ? -1 : 1;
					const int fd = ls[0] + s1 * ls[1] + s2 * ls[2] + s1 * s2 * ls[3];
					cnn = quant_dequant(cw_, 1, p.mfc, p.dvc, f_intra_c, p.qbits_c, p.qshift_c, clv, cd);
					cd[0] = ((fd * p.dvc[0] * 16) << p.qshift_c) >> 5;
					idct4x4_add(cd, cpred, out);
#pragma unroll
					for (int i = 1; i < 16; i++)
						L.coef[COEF_CAC + (4 * c + blk) * 16 + i] = (int16_t)clv[i];
#pragma unroll
					for (int i = 0; i < 16; i++)
						L.c[(1 + 4 * by + (i >> 2)) * IC_STRIDE + 16 + 2 * (4 * bx + (i & 3)) + c] = (uint8_t)out[i];
					L.nnz[lane] = (uint8_t)cnn;
				}
			}
			{
				const uint64_t acm = __ballot(cnn != 0), dcm = __ballot(cdcnz != 0);
				cbp |= acm ? 0x20 : (dcm ? 0x10 : 0);
			}
			WAVE_SYNC();

			// ---- write back ----
			if (lane < COEF_PER_MB * 2 / 16)
				*(u32x4 *)((uint8_t *)(p.coef + mbaddr * COEF_PER_MB) + lane * 16) = *(const u32x4 *)((const uint8_t *)L.coef + lane * 16);
			if (lane < 16)
				*(u32x4 *)(rec_y + (size_t)(y0 + lane) * p.cw + x0) = *(const u32x4 *)(L.t + (1 + lane) * IT_STRIDE + 16);
			else if (lane < 24)
				*(u32x4 *)(rec_uv + (size_t)(my * 8 + lane - 16) * p.cw + x0) = *(const u32x4 *)(L.c + (1 + lane - 16) * IC_STRIDE + 16);
			else if (lane < 48)
				p.nnz[mbaddr * NNZ_PER_MB + lane - 24] = L.nnz[lane - 24];
			if (lane == 0) {
				MbInfo m = mi;
				m.cbp = (int8_t)cbp;
				p.mbi[mbaddr] = m;
			}
			publish_progress(s_prog, my - r0, mx + 1);
		}
	}
}

int Launch::intra_recon(const Params &p, void *stream)
{
	dim3 grid(p.n_streams * p.slice_num);
	hipLaunchKernelGGL(k_intra_recon, grid, dim3(64 * WF_WAVES), 0, (hipStream_t)stream, p);
	return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// k_deblock: each wave sweeps a band of 4 macroblock rows; lane group
// g = lane>>4 works on row band+g, two macroblocks behind group g-1, so all 64
// lanes are busy. Within a group the 16 lanes are the 16 sample rows of the
// macroblock for the vertical edges and its 16 sample columns for the
// horizontal edges; each lane filters its line in registers.
// ------------------------------------------------------------------------
#define DT_STRIDE 32 // luma tile: (x,y) at [(4+y)*32 + 16 + x], x,y in [-4,15]
#define DC_STRIDE 32 // chroma tile (nv12): (x,y,c) at [(2+y)*32 + 16 + 2x + c], x,y in [-2,7]

struct DbGroupLds {
	uint8_t y[20 * DT_STRIDE];
	uint8_t c[10 * DC_STRIDE];
};

DEVI void filt_luma(int bs, int alpha, int beta, int t0, int t1, int t2,
		    int p3, int &p2, int &p1, int &p0, int &q0, int &q1, int &q2, int q3)
{
	if (bs == 0)
		return;
	if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta)
		return;
	const int ap = iabs(p2 - p0), aq = iabs(q2 - q0);
	if (bs < 4) {
		const int tc0 = bs == 1 ? t0 : bs == 2 ? t1 : t2;
		const int tc = tc0 + (ap < beta) + (aq < beta);
		const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
		const int avg = (p0 + q0 + 1) >> 1;
		const int np1 = ap < beta ? p1 + clip3(-tc0, tc0, (p2 + avg - (p1 << 1)) >> 1) : p1;
		const int nq1 = aq < beta ? q1 + clip3(-tc0, tc0, (q2 + avg - (q1 << 1)) >> 1) : q1;
		p0 = clip255(p0 + delta);
		q0 = clip255(q0 - delta);
		p1 = np1;
		q1 = nq1;
	} else {
		const bool small = iabs(p0 - q0) < ((alpha >> 2) + 2);
		int np0, np1 = p1, np2 = p2, nq0, nq1 = q1, nq2 = q2;
		if (ap < beta && small) {
			np0 = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3;
			np1 = (p2 + p1 + p0 + q0 + 2) >> 2;
			np2 = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3;
		} else {
			np0 = (2 * p1 + p0 + q1 + 2) >> 2;
		}
		if (aq < beta && small) {
			nq0 = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3;
			nq1 = (p0 + q0 + q1 + q2 + 2) >> 2;
			nq2 = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3;
		} else {
			nq0 = (2 * q1 + q0 + p1 + 2) >> 2;
		}
		p0 = np0; p1 = np1; p2 = np2;
		q0 = nq0; q1 = nq1; q2 = nq2;
	}
}

DEVI void filt_chroma(int bs, int alpha, int beta, int t0, int t1, int t2, int p1, int &p0, int &q0, int q1)
{
	if (bs == 0)
		return;
	if (iabs(p0 - q0) >= alpha || iabs(p1 - p0) >= beta || iabs(q1 - q0) >= beta)
		return;
	if (bs < 4) {
		const int tc = (bs == 1 ? t0 : bs == 2 ? t1 : t2) + 1;
		const int delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3);
		p0 = clip255(p0 + delta);
		q0 = clip255(q0 - delta);
	} else {
		const int np0 = (2 * p1 + p0 + q1 + 2) >> 2, nq0 = (2 * q1 + q0 + p1 + 2) >> 2;
		p0 = np0;
		q0 = nq0;
	}
}

__global__ __launch_bounds__(64 * WF_WAVES) void k_deblock(Params p)
{
	__shared__ __attribute__((aligned(16))) DbGroupLds s_g[WF_WAVES * 4];
	__shared__ int s_prog[MAX_ROWS];

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int g = lane >> 4, l = lane & 15;
	const int unit = blockIdx.x;
	const int st = unit / p.slice_num, sl = p.slice_first + unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, sl, &r0, &r1);
	for (int i = threadIdx.x; i < MAX_ROWS; i += blockDim.x)
		s_prog[i] = 0;
	__syncthreads();

	DbGroupLds &L = s_g[wave * 4 + g];
	uint8_t *rec_y = p.rec_y + (size_t)st * p.plane_y_stride;
	uint8_t *rec_uv = p.rec_uv + (size_t)st * p.plane_uv_stride;
	const int nbands = (r1 - r0 + 3) >> 2;

	for (int band = wave; band < nbands; band += WF_WAVES) {
		const int my = r0 + 4 * band + g;
		const bool row_ok = my < r1;
		const bool top_edge = my > r0; // picture top, or slice top with idc 2: not filtered
		uint32_t nz_left = 0;
		int mvl = 0;
		bool intra_left = false;

		for (int step = 0; step < p.mb_w + 6; step++) {
			const int mx = step - 2 * g;
			const bool act = row_ok && mx >= 0 && mx < p.mb_w;
			// group 0 depends on the last row of the previous band (another wave)
			if (band > 0) {
				const int mx0 = step; // group 0's column
				if (mx0 < p.mb_w)
					wait_progress(s_prog, 4 * band - 1, mx0 + 2 < p.mb_w ? mx0 + 2 : p.mb_w);
			}
			const int x0 = mx * 16, y0 = my * 16;
			const size_t mbaddr = (size_t)st * p.n_mb + (size_t)(act ? my : r0) * p.mb_w + (act ? mx : 0);
			const bool left_edge = mx > 0;
			MbInfo mi = p.mbi[mbaddr];
			MbInfo mt = mi;
			uint8_t nzb = p.nnz[mbaddr * NNZ_PER_MB + l], nzt_b = 0;
			if (act) {
				// keep the previous macroblock's last 4 columns as this one's left neighbours
				if (left_edge) {
					*(uint32_t *)(L.y + (4 + l) * DT_STRIDE + 12) = *(const uint32_t *)(L.y + (4 + l) * DT_STRIDE + 28);
					if (l < 8)
						*(uint32_t *)(L.c + (2 + l) * DC_STRIDE + 12) = *(const uint32_t *)(L.c + (2 + l) * DC_STRIDE + 28);
				}
				*(u32x4 *)(L.y + (4 + l) * DT_STRIDE + 16) = *(const u32x4 *)(rec_y + (size_t)(y0 + l) * p.cw + x0);
				if (l < 8)
					*(u32x4 *)(L.c + (2 + l) * DC_STRIDE + 16) = *(const u32x4 *)(rec_uv + (size_t)(my * 8 + l) * p.cw + x0);
				if (top_edge) {
					if (l < 4)
						*(u32x4 *)(L.y + l * DT_STRIDE + 16) = *(const u32x4 *)(rec_y + (size_t)(y0 - 4 + l) * p.cw + x0);
					else if (l < 6)
						*(u32x4 *)(L.c + (l - 4) * DC_STRIDE + 16) = *(const u32x4 *)(rec_uv + (size_t)(my * 8 - 2 + l - 4) * p.cw + x0);
					mt = p.mbi[mbaddr - p.mb_w];
					nzt_b = p.nnz[(mbaddr - p.mb_w) * NNZ_PER_MB + l];
				}
			}
			const uint64_t bal_c = __ballot(act && nzb != 0), bal_t = __ballot(act && nzt_b != 0);
			const uint32_t nz_cur = (uint32_t)(bal_c >> (16 * g)) & 0xffffu;
			const uint32_t nz_top = (uint32_t)(bal_t >> (16 * g)) & 0xffffu;
			const bool intra_cur = mi.type <= MB_I16x16, intra_top = mt.type <= MB_I16x16;
			const int mvc = ((int)mi.mvx << 16) ^ ((int)mi.mvy & 0xffff);
			const int mvt = ((int)mt.mvx << 16) ^ ((int)mt.mvy & 0xffff);
			WAVE_SYNC();

			if (act) {
				const int k = l >> 2; // 4-sample segment of this lane's line
				// ---- vertical edges: lane = sample row l ----
				{
					int bs[4];
					bs[0] = !left_edge ? 0
						: (intra_cur || intra_left) ? 4
						: (((nz_cur >> (4 * k)) | (nz_left >> (4 * k + 3))) & 1) ? 2
						: (mvc != mvl) ? 1 : 0;
#pragma unroll
					for (int e = 1; e < 4; e++)
						bs[e] = intra_cur ? 3 : (((nz_cur >> (4 * k + e)) | (nz_cur >> (4 * k + e - 1))) & 1) ? 2 : 0;
					int px[20];
					const uint32_t *rowp = (const uint32_t *)(L.y + (4 + l) * DT_STRIDE + 12);
#pragma unroll
					for (int w = 0; w < 5; w++) {
						uint32_t v = rowp[w];
						px[4 * w] = v & 0xff;
						px[4 * w + 1] = (v >> 8) & 0xff;
						px[4 * w + 2] = (v >> 16) & 0xff;
						px[4 * w + 3] = v >> 24;
					}
#pragma unroll
					for (int e = 0; e < 4; e++) {
						const int q = 4 + 4 * e;
						filt_luma(bs[e], p.alpha_y, p.beta_y, p.tc0_y[0], p.tc0_y[1], p.tc0_y[2],
							  px[q - 4], px[q - 3], px[q - 2], px[q - 1], px[q], px[q + 1], px[q + 2], px[q + 3]);
					}
					uint32_t *roww = (uint32_t *)(L.y + (4 + l) * DT_STRIDE + 12);
#pragma unroll
					for (int w = 0; w < 5; w++)
						roww[w] = (uint32_t)px[4 * w] | ((uint32_t)px[4 * w + 1] << 8) | ((uint32_t)px[4 * w + 2] << 16) | ((uint32_t)px[4 * w + 3] << 24);
					// chroma: lane = (plane c = l>>3, chroma row r = l&7); edges at chroma x = 0 and 4
					{
						const int c = l >> 3, r = l & 7, ks = r >> 1;
						int bsc0 = !left_edge ? 0
							: (intra_cur || intra_left) ? 4
							: (((nz_cur >> (4 * ks)) | (nz_left >> (4 * ks + 3))) & 1) ? 2
							: (mvc != mvl) ? 1 : 0;
						int bsc2 = intra_cur ? 3 : (((nz_cur >> (4 * ks + 2)) | (nz_cur >> (4 * ks + 1))) & 1) ? 2 : 0;
						uint8_t *cp = L.c + (2 + r) * DC_STRIDE + 16 + c;
						int a1 = cp[-4], a0 = cp[-2], b0 = cp[0], b1 = cp[2];
						filt_chroma(bsc0, p.alpha_c, p.beta_c, p.tc0_c[0], p.tc0_c[1], p.tc0_c[2], a1, a0, b0, b1);
						cp[-2] = (uint8_t)a0;
						cp[0] = (uint8_t)b0;
						int c1 = cp[4], c0 = cp[6], d0 = cp[8], d1 = cp[10];
						filt_chroma(bsc2, p.alpha_c, p.beta_c, p.tc0_c[0], p.tc0_c[1], p.tc0_c[2], c1, c0, d0, d1);
						cp[6] = (uint8_t)c0;
						cp[8] = (uint8_t)d0;
					}
				}
			}
			WAVE_SYNC();
			if (act) {
				const int k = l >> 2;
				// ---- horizontal edges: lane = sample column l ----
				{
					int bs[4];
					bs[0] = !top_edge ? 0
						: (intra_cur || intra_top) ? 4
						: (((nz_cur >> k) | (nz_top >> (12 + k))) & 1) ? 2
						: (mvc != mvt) ? 1 : 0;
#pragma unroll
					for (int e = 1; e < 4; e++)
						bs[e] = intra_cur ? 3 : (((nz_cur >> (4 * e + k)) | (nz_cur >> (4 * e - 4 + k))) & 1) ? 2 : 0;
					int py[20];
					uint8_t *colp = L.y + 16 + l;
#pragma unroll
					for (int r = 0; r < 20; r++)
						py[r] = colp[r * DT_STRIDE];
#pragma unroll
					for (int e = 0; e < 4; e++) {
						const int q = 4 + 4 * e;
						filt_luma(bs[e], p.alpha_y, p.beta_y, p.tc0_y[0], p.tc0_y[1], p.tc0_y[2],
							  py[q - 4], py[q - 3], py[q - 2], py[q - 1], py[q], py[q + 1], py[q + 2], py[q + 3]);
					}
#pragma unroll
					for (int r = 1; r < 19; r++)
						colp[r * DT_STRIDE] = (uint8_t)py[r];
					// chroma: lane = byte column l of the nv12 row (plane l&1, chroma x = l>>1)
					{
						int bsc0 = bs[0];
						int bsc2 = bs[2];
						uint8_t *cp = L.c + 16 + l;
						int a1 = cp[0], a0 = cp[DC_STRIDE], b0 = cp[2 * DC_STRIDE], b1 = cp[3 * DC_STRIDE];
						filt_chroma(bsc0, p.alpha_c, p.beta_c, p.tc0_c[0], p.tc0_c[1], p.tc0_c[2], a1, a0, b0, b1);
						cp[DC_STRIDE] = (uint8_t)a0;
						cp[2 * DC_STRIDE] = (uint8_t)b0;
						int c1 = cp[4 * DC_STRIDE], c0 = cp[5 * DC_STRIDE], d0 = cp[6 * DC_STRIDE], d1 = cp[7 * DC_STRIDE];
						filt_chroma(bsc2, p.alpha_c, p.beta_c, p.tc0_c[0], p.tc0_c[1], p.tc0_c[2], c1, c0, d0, d1);
						cp[5 * DC_STRIDE] = (uint8_t)c0;
						cp[6 * DC_STRIDE] = (uint8_t)d0;
					}
				}
			}
			WAVE_SYNC();
			if (act) {
				// ---- write back: interior, 4 left columns, 4 top rows ----
				*(u32x4 *)(rec_y + (size_t)(y0 + l) * p.cw + x0) = *(const u32x4 *)(L.y + (4 + l) * DT_STRIDE + 16);
				if (left_edge)
					*(uint32_t *)(rec_y + (size_t)(y0 + l) * p.cw + x0 - 4) = *(const uint32_t *)(L.y + (4 + l) * DT_STRIDE + 12);
				if (l < 8) {
					*(u32x4 *)(rec_uv + (size_t)(my * 8 + l) * p.cw + x0) = *(const u32x4 *)(L.c + (2 + l) * DC_STRIDE + 16);
					if (left_edge)
						*(uint32_t *)(rec_uv + (size_t)(my * 8 + l) * p.cw + x0 - 4) = *(const uint32_t *)(L.c + (2 + l) * DC_STRIDE + 12);
				}
				if (top_edge) {
					if (l < 4)
						*(u32x4 *)(rec_y + (size_t)(y0 - 4 + l) * p.cw + x0) = *(const u32x4 *)(L.y + l * DT_STRIDE + 16);
					else if (l < 6)
						*(u32x4 *)(rec_uv + (size_t)(my * 8 - 2 + l - 4) * p.cw + x0) = *(const u32x4 *)(L.c + (l - 4) * DC_STRIDE + 16);
				}
				nz_left = nz_cur;
				mvl = mvc;
				intra_left = intra_cur;
			}
			// publish: one lane per group, after this wave's stores have completed
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
			if (act && l == 0)
				__hip_atomic_store(&s_prog[my - r0], mx + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		}
	}
}

int Launch::deblock(const Params &p, void *stream)
{
	dim3 grid(p.n_streams * p.slice_num);
	hipLaunchKernelGGL(k_deblock, grid, dim3(64 * WF_WAVES), 0, (hipStream_t)stream, p);
	return (int)hipGetLastError();
}

} // namespace cedar
