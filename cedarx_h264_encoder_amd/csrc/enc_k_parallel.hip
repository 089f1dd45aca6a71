// cedarx_h264_encoder_amd/csrc/enc_k_parallel.hip
// The macroblock-parallel gfx950 kernels: input padding, open-loop intra mode
// decision, and the fused P-macroblock kernel (16x16 SAD search against the
// deblocked reference + residual + 4x4 transform/quant + reconstruction).
// One 64-lane wavefront per macroblock; tiles are staged in LDS from 16-byte
// coalesced loads; SAD candidates are evaluated with v_qsad_pk_u16_u8 (four
// horizontal offsets per instruction) and reduced across the wave with
// shuffles. No MFMA: there is no dense contraction on this path.
//
// These replace the VE's ISP fetch + ME + transform stages that the reference
// only triggers (kernel/cedar.c:1068-1080, :1142-1149, :1170-1176).
#include <hip/hip_runtime.h>

#include <time.h>
#include "enc_device.h"

namespace cedar {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// XCD-aware block order. Workgroups are dealt round-robin over the 8 XCDs (each
// with its own 4 MiB L2), so neighbouring macroblocks, whose reference windows
// overlap by 8/9, would land on different L2s and every window would be fetched
// from the memory side (measured: 8x the algorithmic bytes). Renumber so that the
// blocks sharing an XCD (equal id % 8) own one contiguous run of macroblocks.
// Bijective for any block count (guide: T1). Speed only, never correctness.
DEVI void xcd_mb(const Params &p, int &mx, int &my, int &st)
{
	const uint32_t nb = gridDim.x, hb = blockIdx.x;
	const uint32_t xcd = hb & 7u, i = hb >> 3, q = nb >> 3, r = nb & 7u;
	const uint32_t lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + i;
	const uint32_t per_stream = (uint32_t)(p.mb_w * p.row_num);
	st = (int)(lb / per_stream);
	const uint32_t m = lb - (uint32_t)st * per_stream;
	my = p.row_first + (int)(m / (uint32_t)p.mb_w);
	mx = (int)(m % (uint32_t)p.mb_w);
}

// ------------------------------------------------------------------------
// k_input: tightly packed width x height nv12 -> padded coded-size planes
// (edge replication right/bottom). One thread per 16 output bytes.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_input(Params p)
{
	const int segs = p.cw >> 4;
	const int seg = blockIdx.x * blockDim.x + threadIdx.x;
	// only the macroblock rows this session codes (a slice session on one of several GPUs reads
	// nothing outside its band: intra prediction and the motion search stop at the slice)
	const int band_y = 16 * p.row_num;
	const int row = (int)blockIdx.y < band_y ? 16 * p.row_first + (int)blockIdx.y
						 : p.ch + 8 * p.row_first + ((int)blockIdx.y - band_y); // [0, ch) luma, [ch, ch + ch/2) chroma
	const int st = blockIdx.z;
	// first job of every frame on this stream: the ticket / abort / progress words of the intra
	// wavefront kernel (region 0) start at zero; the deblock regions are cleared on the deblock
	// kernel's own stream
	if (blockIdx.y == 0 && st == 0 && p.wf_split_intra > 1) {
		const int used = 16 + p.n_streams * p.slice_num * WF_MAX_SPLIT;
		for (int i = seg; i < used; i += gridDim.x * blockDim.x)
			p.wf_sync[i] = 0;
	}
	if (seg >= segs)
		return;
	const bool luma = row < p.ch;
	const int prow = luma ? row : row - p.ch;
	const int srows = luma ? p.height : p.height >> 1;
	const int sy = prow < srows ? prow : srows - 1;
	const uint8_t *base;
	if (p.in_luma) {
		base = luma ? p.in_luma[st] : p.in_chroma[st];
	} else { // wave-uniform pick from the kernel arguments without indexing them dynamically
		base = nullptr;
#pragma unroll
		for (int k = 0; k < IN_K_STREAMS; k++)
			base = st == k ? (luma ? p.in_k[k] : p.in_k[IN_K_STREAMS + k]) : base;
	}
	const uint8_t *src = base + (size_t)sy * p.width;
	uint8_t *dst = luma ? p.src_y + (size_t)st * p.plane_y_stride + (size_t)prow * p.cw
			    : p.src_uv + (size_t)st * p.plane_uv_stride + (size_t)prow * p.cw;
	const int x0 = seg << 4;
	u32x4 v;
	if (x0 + 16 <= p.width && ((p.width & 15) == 0) && ((((uintptr_t)src) & 15) == 0)) {
		v = *(const u32x4 *)(src + x0);
	} else {
		uint32_t w[4] = { 0, 0, 0, 0 };
#pragma unroll
		for (int i = 0; i < 16; i++) {
			int x = x0 + i, sx;
			if (luma)
				sx = x < p.width ? x : p.width - 1;
			else { // replicate the last CbCr pair
				int pr = x >> 1, lim = (p.width >> 1) - 1;
				sx = 2 * (pr < lim ? pr : lim) + (x & 1);
			}
			w[i >> 2] |= (uint32_t)src[sx] << (8 * (i & 3));
		}
		v = u32x4{ w[0], w[1], w[2], w[3] };
	}
	*(u32x4 *)(dst + x0) = v;
}

int Launch::input(const Params &p, void *stream)
{
	dim3 grid((p.cw / 16 + 255) / 256, 24 * p.row_num, p.n_streams);
	hipLaunchKernelGGL(k_input, grid, dim3(256), 0, (hipStream_t)stream, p);
	return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// k_inter: one wave per macroblock of a P picture.
// LDS: reference window 48x48 luma (+-16), 25 rows x 32 CbCr pairs chroma.
// ------------------------------------------------------------------------
#define REF_STRIDE 48
#define REFC_STRIDE 64

// k_inter: macroblock number i of this launch -> (mx, my, st). A launch covers the whole
// picture (stripe_s1 <= 0) or the diagonals stripe_s0 <= x + 2*y < stripe_s1 of every stream,
// enumerated row by row without holes (stripe_tab = running count per row). False past the end.
DEVI bool inter_mb(const Params &p, uint32_t i, int &mx, int &my, int &st)
{
	if (p.stripe_s1 <= 0) {
		const uint32_t per_stream = (uint32_t)(p.mb_w * p.row_num);
		st = (int)(i / per_stream);
		const uint32_t m = i - (uint32_t)st * per_stream;
		my = p.row_first + (int)(m / (uint32_t)p.mb_w);
		mx = (int)(m % (uint32_t)p.mb_w);
		return st < p.n_streams;
	}
	const uint32_t per_stream = p.stripe_tab[p.row_num];
	st = (int)(i / per_stream);
	const uint32_t m = i - (uint32_t)st * per_stream;
	int lo = 0, hi = p.row_num; // last row r with tab[r] <= m
	while (hi - lo > 1) {
		const int mid = (lo + hi) >> 1;
		if (p.stripe_tab[mid] <= m)
			lo = mid;
		else
			hi = mid;
	}
	my = p.row_first + lo;
	const int xs = p.stripe_s0 - 2 * my;
	mx = (xs > 0 ? xs : 0) + (int)(m - p.stripe_tab[lo]);
	return st < p.n_streams;
}

__global__ __launch_bounds__(64 * INTER_WAVES) void k_inter(Params p)
{
	// one wave per macroblock, INTER_WAVES macroblocks per workgroup (each wave has its own
	// slice of every array; the barriers below only ever order a wave with itself)
	__shared__ __attribute__((aligned(16))) uint8_t sa_ref[INTER_WAVES][48 * REF_STRIDE + 64];
	__shared__ __attribute__((aligned(16))) uint8_t sa_refc[INTER_WAVES][25 * REFC_STRIDE];
	__shared__ __attribute__((aligned(16))) uint8_t sa_cur[INTER_WAVES][256];
	__shared__ __attribute__((aligned(16))) uint8_t sa_curc[INTER_WAVES][128];
	__shared__ __attribute__((aligned(16))) int16_t sa_coef[INTER_WAVES][COEF_PER_MB];
	__shared__ __attribute__((aligned(16))) uint8_t sa_recy[INTER_WAVES][256];
	__shared__ __attribute__((aligned(16))) uint8_t sa_recc[INTER_WAVES][128];
	__shared__ int sa_dc[INTER_WAVES][8];
	__shared__ int sa_cdc[INTER_WAVES][8];

	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	uint8_t *const s_ref = sa_ref[wave], *const s_refc = sa_refc[wave], *const s_cur = sa_cur[wave], *const s_curc = sa_curc[wave];
	int16_t *const s_coef = sa_coef[wave];
	uint8_t *const s_recy = sa_recy[wave], *const s_recc = sa_recc[wave];
	int *const s_dc = sa_dc[wave], *const s_cdc = sa_cdc[wave];
	// XCD-aware workgroup order (see xcd_mb): workgroups sharing an XCD own one contiguous run
	const uint32_t nb = gridDim.x, hb = blockIdx.x;
	const uint32_t xcd = hb & 7u, bi = hb >> 3, q = nb >> 3, r = nb & 7u;
	const uint32_t lb = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + bi;
	int mx, my, st;
	// an idle wave of the last workgroup redoes macroblock 0 and writes to the dummy slots
	const bool valid = inter_mb(p, lb * INTER_WAVES + (uint32_t)wave, mx, my, st);
	if (!valid) {
		mx = 0;
		my = p.row_first;
		st = 0;
	}
	const int x0 = mx * 16, y0 = my * 16;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, slice_of_row(p.mb_h, p.slice_count, my), &r0, &r1);
	const int band_y0 = r0 * 16, band_y1 = r1 * 16;

	const uint8_t *src_y = p.src_y + (size_t)st * p.plane_y_stride;
	const uint8_t *src_uv = p.src_uv + (size_t)st * p.plane_uv_stride;
	const uint8_t *ref_y = p.ref_y + (size_t)st * p.plane_y_stride;
	const uint8_t *ref_uv = p.ref_uv + (size_t)st * p.plane_uv_stride;
	uint8_t *rec_y = valid ? p.rec_y + (size_t)st * p.plane_y_stride + (size_t)y0 * p.cw + x0 : p.dummy_rec;
	uint8_t *rec_uv = valid ? p.rec_uv + (size_t)st * p.plane_uv_stride + (size_t)(my * 8) * p.cw + x0 : p.dummy_rec + (size_t)16 * p.cw;
	// records, counts and levels of the dummy macroblock live one past the last stream
	const size_t mbaddr = valid ? (size_t)st * p.n_mb + (size_t)my * p.mb_w + mx : (size_t)p.dummy_mb;

	// ---- stage tiles through LDS with 16-byte loads ----
	if (lane < 16)
		*(u32x4 *)(s_cur + lane * 16) = *(const u32x4 *)(src_y + (size_t)(y0 + lane) * p.cw + x0);
	else if (lane < 24)
		*(u32x4 *)(s_curc + (lane - 16) * 16) =
			*(const u32x4 *)(src_uv + (size_t)(my * 8 + lane - 16) * p.cw + x0);
	for (int idx = lane; idx < 48 * 3; idx += 64) {
		int row = idx / 3, seg = idx - row * 3;
		int gy = y0 - 16 + row, gx = x0 - 16 + 16 * seg;
		u32x4 v = u32x4{ 0, 0, 0, 0 };
		if (gy >= 0 && gy < p.ch && gx >= 0 && gx < p.cw)
			v = *(const u32x4 *)(ref_y + (size_t)gy * p.cw + gx);
		*(u32x4 *)(s_ref + row * REF_STRIDE + 16 * seg) = v;
	}
	for (int idx = lane; idx < 25 * 4; idx += 64) {
		int row = idx >> 2, seg = idx & 3;
		int gy = my * 8 - 8 + row, gx = x0 - 16 + 16 * seg; // byte offset in the CbCr row
		u32x4 v = u32x4{ 0, 0, 0, 0 };
		if (gy >= 0 && gy < (p.ch >> 1) && gx >= 0 && gx < p.cw)
			v = *(const u32x4 *)(ref_uv + (size_t)gy * p.cw + gx);
		*(u32x4 *)(s_refc + row * REFC_STRIDE + 16 * seg) = v;
	}
	if (lane < 16)
		*(u32x4 *)(s_ref + 48 * REF_STRIDE + (lane & 3) * 16) = u32x4{ 0, 0, 0, 0 };
	__syncthreads();

	// ---- motion search: lane task = (dy, group of 4 dx); 33 x 9 = 297 tasks ----
	uint32_t curd[64];
#pragma unroll
	for (int i = 0; i < 64; i++)
		curd[i] = ((const uint32_t *)s_cur)[i];

	// lane task: dx group g = lane % 9 (fixed for the lane, so its 4 dx costs and their
	// validity are computed once), dy = 7 * round + lane / 9 - 16; lane 63 idles.
	const int R = p.search_range;
	uint32_t best = 0xffffffffu;
	const int g9 = lane % 9, dsub = lane / 9;
	const int dx0 = 4 * g9 - 16;
	uint32_t cost_x[4];
	bool ok_x[4];
#pragma unroll
	for (int k = 0; k < 4; k++) {
		const int dx = dx0 + k;
		ok_x[k] = lane < 63 && dx >= -R && dx <= R && x0 + dx >= 0 && x0 + dx + 16 <= p.cw;
		cost_x[k] = (uint32_t)(p.lambda * mv_bits(dx)) << 11 | (uint32_t)(dx + 16);
	}
	for (int round = 0; round < 5; round++) {
		const int dyi_raw = 7 * round + dsub;
		const bool tvalid = dyi_raw < 33;
		const int dyi = tvalid ? dyi_raw : 32;
		const int dy = dyi - 16;
		const uint8_t *base = s_ref + dyi * REF_STRIDE + 4 * g9;
		// four independent accumulators (one per 4-sample segment) keep the qsad
		// pipeline full; each lane's 4 x u16 partial SADs are summed afterwards
		uint64_t a0 = 0, a1 = 0, a2 = 0, a3 = 0;
#pragma unroll
		for (int r = 0; r < 16; r++) {
			const uint32_t *rp = (const uint32_t *)(base + r * REF_STRIDE);
			uint32_t w0 = rp[0], w1 = rp[1], w2 = rp[2], w3 = rp[3], w4 = rp[4];
			a0 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w1 << 32) | w0, curd[4 * r + 0], a0);
			a1 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w2 << 32) | w1, curd[4 * r + 1], a1);
			a2 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w3 << 32) | w2, curd[4 * r + 2], a2);
			a3 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w4 << 32) | w3, curd[4 * r + 3], a3);
		}
		// per-u16 sums cannot carry: a full 16x16 SAD is at most 65280
		const uint64_t acc = (a0 + a1) + (a2 + a3);
		const bool yok = tvalid && dy >= -R && dy <= R && y0 + dy >= band_y0 && y0 + dy + 16 <= band_y1;
		// key = cost << 11 | candidate index, index = (dy+16)*33 + dx+16
		const uint32_t key_y = ((uint32_t)(p.lambda * mv_bits(dy)) << 11) + (uint32_t)((dy + 16) * 33);
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const uint32_t sad = (uint32_t)(acc >> (16 * k)) & 0xffffu;
			const uint32_t key = (sad << 11) + key_y + cost_x[k];
			best = (yok && ok_x[k] && key < best) ? key : best;
		}
	}
	best = wave_min_u32(best);
	const int bidx = (int)(best & 2047u);
	const int bdy = bidx / 33 - 16, bdx = bidx - (bidx / 33) * 33 - 16;

	// ---- residual, transform, quantisation, reconstruction: one lane per SAMPLE ----
	// 16-lane group = one 4x4 block, transforms across lanes with DPP (enc_device.h):
	// 4 passes cover the 16 luma blocks, 2 passes the 8 chroma blocks, all 64 lanes busy.
	for (int i = lane; i < COEF_PER_MB; i += 64)
		s_coef[i] = 0;
	__syncthreads();

	const int f_inter = (1 << p.qbits) / 6, f_inter_c = (1 << p.qbits_c) / 6;
	const int gq = lane >> 4, l = lane & 15;
	const int pi = l >> 2, pj = l & 3;
	const LaneK ci = lane_consts(pi), cj = lane_consts(pj);
	const int kv = pi == 1 ? 2 : pi == 2 ? 1 : pi, kh = pj == 1 ? 2 : pj == 2 ? 1 : pj;
	const int rpos = 4 * kv + kh;             // coefficient (raster) this lane ends up holding
	const int pclass = (kv & 1) == (kh & 1) ? (kv & 1) : 2;
	int zz = 0;
#pragma unroll
	for (int z = 0; z < 16; z++)
		zz = kZigzag[z] == rpos ? z : zz;
	const int mf_y = pclass == 0 ? p.mf[0] : pclass == 1 ? p.mf[1] : p.mf[2];
	const int dv_y = pclass == 0 ? p.dv[0] : pclass == 1 ? p.dv[1] : p.dv[2];
	const int mf_c = pclass == 0 ? p.mfc[0] : pclass == 1 ? p.mfc[1] : p.mfc[2];
	const int dv_c = pclass == 0 ? p.dvc[0] : pclass == 1 ? p.dvc[1] : p.dvc[2];

	uint32_t nz_luma = 0; // bit = coding-order block index
#pragma unroll
	for (int pass = 0; pass < 4; pass++) {
		const int idx = 4 * pass + gq, bx = blk_x(idx), by = blk_y(idx);
		const int x = 4 * bx + pj, y = 4 * by + pi;
		const int pv = s_ref[(16 + bdy + y) * REF_STRIDE + 16 + bdx + x];
		const int res = (int)s_cur[y * 16 + x] - pv;
		const int w = fdct_lanes(res, ci, cj);
		const int level = quant1(w, mf_y, f_inter, p.qbits);
		s_coef[COEF_LUMA + idx * 16 + zz] = (int16_t)level;
		const int r = idct_lanes((level * dv_y) << p.qshift, ci, cj);
		s_recy[y * 16 + x] = (uint8_t)clip255(pv + r);
		const uint64_t bal = __ballot(level != 0);
		const int nn = __popc((uint32_t)(bal >> (16 * gq)) & 0xffffu);
		if (l == 0)
			p.nnz[mbaddr * NNZ_PER_MB + bx + 4 * by] = (uint8_t)nn;
#pragma unroll
		for (int k = 0; k < 4; k++)
			nz_luma |= ((uint32_t)(bal >> (16 * k)) & 0xffffu) ? (1u << (4 * pass + k)) : 0u;
	}
	// chroma: pass = plane, group = block (raster 2x2)
	const int ix = bdx >> 1, iy = bdy >> 1, fx = (bdx & 1) * 4, fy = (bdy & 1) * 4;
	int cpv[2], cdq[2];
	uint32_t nz_cac = 0;
#pragma unroll
	for (int c = 0; c < 2; c++) {
		const int bx = gq & 1, by = gq >> 1;
		const int x = 4 * bx + pj, y = 4 * by + pi;
		const int wr = y + iy + 8, wc = x + ix + 8; // window row / pair index
		const int wr1 = fy ? wr + 1 : wr, wc1 = fx ? wc + 1 : wc;
		const int A = s_refc[wr * REFC_STRIDE + 2 * wc + c], B = s_refc[wr * REFC_STRIDE + 2 * wc1 + c];
		const int C = s_refc[wr1 * REFC_STRIDE + 2 * wc + c], D = s_refc[wr1 * REFC_STRIDE + 2 * wc1 + c];
		cpv[c] = ((8 - fx) * (8 - fy) * A + fx * (8 - fy) * B + (8 - fx) * fy * C + fx * fy * D + 32) >> 6;
		const int res = (int)s_curc[y * 16 + 2 * x + c] - cpv[c];
		const int w = fdct_lanes(res, ci, cj);
		if (l == 0)
			s_dc[4 * c + gq] = w; // lane (0,0) holds the DC coefficient
		const int level = l == 0 ? 0 : quant1(w, mf_c, f_inter_c, p.qbits_c);
		if (l != 0)
			s_coef[COEF_CAC + (4 * c + gq) * 16 + zz] = (int16_t)level;
		cdq[c] = (level * dv_c) << p.qshift_c;
		const uint64_t bal = __ballot(level != 0);
		const int nn = __popc((uint32_t)(bal >> (16 * gq)) & 0xffffu);
		if (l == 0)
			p.nnz[mbaddr * NNZ_PER_MB + 16 + 4 * c + gq] = (uint8_t)nn;
		nz_cac |= bal ? 1u : 0u;
	}
	__syncthreads();
	int dc_level_nz = 0;
	if (lane < 8) { // 2x2 Hadamard of the chroma DCs: lane = (plane, output index)
		const int c = lane >> 2, blk = lane & 3;
		const int *dcs = s_dc + 4 * c;
		const int s1 = (blk & 1) ? -1 : 1, s2 = (blk & 2) ? -1 : 1;
		const int hd = dcs[0] + s1 * dcs[1] + s2 * dcs[2] + s1 * s2 * dcs[3];
		const int lq = quant1(hd, p.mfc[0], 2 * f_inter_c, p.qbits_c + 1);
		s_cdc[lane] = lq;
		s_coef[COEF_CDC + lane] = (int16_t)lq;
		dc_level_nz = lq != 0;
	}
	__syncthreads();
#pragma unroll
	for (int c = 0; c < 2; c++) {
		const int bx = gq & 1, by = gq >> 1;
		const int x = 4 * bx + pj, y = 4 * by + pi;
		const int *ls = s_cdc + 4 * c;
		const int s1 = (gq & 1) ? -1 : 1, s2 = (gq & 2) ? -1 : 1;
		const int fd = ls[0] + s1 * ls[1] + s2 * ls[2] + s1 * s2 * ls[3];
		const int dcr = ((fd * p.dvc[0] * 16) << p.qshift_c) >> 5;
		const int r = idct_lanes(l == 0 ? dcr : cdq[c], ci, cj);
		s_recc[y * 16 + 2 * x + c] = (uint8_t)clip255(cpv[c] + r);
	}
	// coded block pattern
	const uint64_t dcmask = __ballot(dc_level_nz != 0);
	int cbp = 0;
#pragma unroll
	for (int q = 0; q < 4; q++)
		cbp |= ((nz_luma >> (4 * q)) & 0xfu) ? (1 << q) : 0;
	cbp |= nz_cac ? 0x20 : (dcmask ? 0x10 : 0);
	__syncthreads();

	// ---- write back: levels (51 x 16 B), reconstruction rows, records ----
	if (lane < COEF_PER_MB * 2 / 16)
		*(u32x4 *)((uint8_t *)(p.coef + mbaddr * COEF_PER_MB) + lane * 16) = *(const u32x4 *)((const uint8_t *)s_coef + lane * 16);
	if (lane < 16)
		*(u32x4 *)(rec_y + (size_t)lane * p.cw) = *(const u32x4 *)(s_recy + lane * 16);
	else if (lane < 24)
		*(u32x4 *)(rec_uv + (size_t)(lane - 16) * p.cw) = *(const u32x4 *)(s_recc + (lane - 16) * 16);
	if (lane < 16)
		p.i4modes[mbaddr * 16 + lane] = 2;
	if (lane == 0) {
		MbInfo m;
		m.type = MB_P16x16;
		m.i16_mode = 0;
		m.chroma_mode = 0;
		m.cbp = (int8_t)cbp;
		m.mvx = (int16_t)bdx;
		m.mvy = (int16_t)bdy;
		p.mbi[mbaddr] = m;
	}
}

int Launch::inter(const Params &p, void *stream, int n_mbs)
{
	hipLaunchKernelGGL(k_inter, dim3((n_mbs + INTER_WAVES - 1) / INTER_WAVES), dim3(64 * INTER_WAVES), 0, (hipStream_t)stream, p);
	return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// k_gate: the only place a frame WAITS for the frame before it. One wave. A P picture is
// analysed in a few diagonal stripes; in front of stripe [s0, s1) this kernel (1) publishes
// how far this frame's analysis has come (the deblock kernel of this frame, already running
// on another stream, polls that word) and (2) spins until every deblock band of the previous
// frame has written back the macroblocks with x + 2*y < need_d: the motion search of (x, y)
// reads the final samples of (x-1..x+1, y-1..y+1), final once their right and lower
// neighbours are filtered, i.e. need_d = s1 + 5.
// Spins are bounded: a time-out leaves an error code and lets the stream drain.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gate(Params p, int publish_d, int need_d)
{
	const int lane = threadIdx.x;
	if (lane == 0 && publish_d >= 0 && p.prog_inter) // the launches before this one have completed (stream order)
		__hip_atomic_store(p.prog_inter, p.prog_tag | (uint32_t)publish_d, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	if (need_d <= 0 || !p.prev_db)
		return;
	const int n_units = p.n_streams * p.slice_num;
	const int n = 2 * n_units * PROG_BANDS;
	for (int e0 = 0; e0 < n; e0 += 64) {
		const int e = e0 + lane;
		bool pending = false;
		if (e < n) {
			const int band = e % PROG_BANDS, unit = (e / PROG_BANDS) % n_units;
			int r0, r1;
			slice_rows(p.mb_h, p.slice_count, p.slice_first + unit % p.slice_num, &r0, &r1);
			// a band owns the diagonals from 2*(r0 + 4*band) on: nothing of it is below that
			pending = band < ((r1 - r0 + 3) >> 2) && 2 * (r0 + 4 * band) < need_d;
		}
		uint32_t spins = 0;
		while (true) {
			if (pending) {
				const uint32_t v = __hip_atomic_load(&p.prev_db[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				pending = (int32_t)(v - (p.prev_tag | (uint32_t)need_d)) < 0;
			}
			if (!__any(pending))
				break;
			if (++spins > (1u << 20)) {
				if (lane == 0)
					__hip_atomic_store(p.wf_err, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				return;
			}
			__builtin_amdgcn_s_sleep(32);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// k_copy_out: the frame's bitstreams (compact, all streams back to back) and their sizes go
// to the caller's pinned buffers by plain coalesced stores; host_out / host_sizes are device
// views of mapped host memory. hipMemcpyAsync did this before: its staging stalled a submit
// for ~7 ms three times while a pipeline of frames warmed up, and the size had to be guessed.
static __device__ __forceinline__ void xstore_u32(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ __launch_bounds__(256) void k_copy_out(Params p, uint8_t *host_out, uint32_t *host_sizes)
{
	const uint32_t total = p.stream_size[p.n_streams];
	const uint32_t n16 = ((total < (uint32_t)p.out_cap ? total : (uint32_t)p.out_cap) + 15u) >> 4;
	for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += gridDim.x * blockDim.x)
		((u32x4 *)host_out)[i] = ((const u32x4 *)p.out)[i];
	if (blockIdx.x == 0 && (int)threadIdx.x < p.n_streams + 2)
		host_sizes[threadIdx.x] = p.stream_size[threadIdx.x];
	if (!p.done_flag)
		return;
	// Pipelined frames: the last workgroup raises the frame's "bitstream delivered" flag (after
	// every workgroup's stores to host memory have been performed) and leaves the size words
	// zeroed for the lane's next frame: no k_flag, no memset on the entropy stream.
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
	__syncthreads();
	if (threadIdx.x == 0) {
		uint32_t *cnt = p.stream_size + p.n_streams + 3;
		if (__hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
			for (int i = 0; i < p.n_streams + 4; i++)
				xstore_u32(&p.stream_size[i], 0u);
			__hip_atomic_store(p.done_flag, p.done_tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
		}
	}
}

int Launch::copy_out(const Params &p, void *stream, uint8_t *host_out, uint32_t *host_sizes)
{
	hipLaunchKernelGGL(k_copy_out, dim3(p.n_streams > 4 ? 256 : 64), dim3(256), 0, (hipStream_t)stream, p, host_out, host_sizes);
	return (int)hipGetLastError();
}

__global__ void k_flag(uint32_t *word, uint32_t value)
{
	__hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

int Launch::flag(void *stream, uint32_t *word, uint32_t value)
{
	hipLaunchKernelGGL(k_flag, dim3(1), dim3(1), 0, (hipStream_t)stream, word, value);
	return (int)hipGetLastError();
}

__global__ void k_probe_wait(uint32_t *w)
{
	// bounded: ~20 ms at most if the setter never runs beside us
	for (int i = 0; i < (1 << 15); i++) {
		if (__hip_atomic_load(&w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
			w[1] = 1;
			return;
		}
		__builtin_amdgcn_s_sleep(64);
	}
	w[1] = 0;
}

__global__ void k_probe_set(uint32_t *w)
{
	__hip_atomic_store(&w[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

int Launch::probe_overlap(void *waiter, void *setter, uint32_t *d_words, bool *overlaps)
{
	// A "no" is tried again: on a box that is still paging things in, the setter's launch can take
	// longer than the waiter's patience once, and a false "no" costs the whole frame pipelining
	// (two streams that do share a hardware queue say "no" every time).
	*overlaps = false;
	for (int attempt = 0; attempt < 3 && !*overlaps; attempt++) {
		uint32_t h[2] = { 0, 0 };
		if (hipMemcpy(d_words, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess)
			return -1;
		hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, (hipStream_t)waiter, d_words);
		hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, (hipStream_t)setter, d_words);
		if (hipStreamSynchronize((hipStream_t)waiter) != hipSuccess || hipStreamSynchronize((hipStream_t)setter) != hipSuccess)
			return -1;
		if (hipMemcpy(h, d_words, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
			return -1;
		*overlaps = h[1] != 0;
	}
	return 0;
}

__global__ void k_probe_nop(uint32_t *w)
{
	if (threadIdx.x == 1)
		w[2] = 1;
}

// Hardware queues come in groups that share a command-processor pipe (two of eight per pipe on
// MI355X with GPU_MAX_HW_QUEUES=8). A pipe whose one queue holds a running kernel WITH a packet
// queued behind it serves its other queue slowly: ~25 us per kernel launch instead of ~3 us
// (measured, tools/dev/pipe_probe.hip). This times a chain of tiny kernels on `chain` while
// `blocked` is in that state; the caller compares with the unblocked time.
int Launch::probe_pipe(void *blocked, void *chain, uint32_t *d_words, int n_chain, double *us_per_kernel)
{
	uint32_t h[4] = { 0, 0, 0, 0 };
	if (hipMemcpy(d_words, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess)
		return -1;
	if (blocked) {
		hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(1), 0, (hipStream_t)blocked, d_words);
		hipLaunchKernelGGL(k_probe_nop, dim3(1), dim3(64), 0, (hipStream_t)blocked, d_words);
	}
	timespec a, b;
	clock_gettime(CLOCK_MONOTONIC, &a);
	for (int i = 0; i < n_chain; i++)
		hipLaunchKernelGGL(k_probe_nop, dim3(1), dim3(64), 0, (hipStream_t)chain, d_words);
	if (hipStreamSynchronize((hipStream_t)chain) != hipSuccess)
		return -1;
	clock_gettime(CLOCK_MONOTONIC, &b);
	*us_per_kernel = ((b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3) / n_chain;
	if (blocked) {
		hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(1), 0, (hipStream_t)chain, d_words);
		if (hipStreamSynchronize((hipStream_t)chain) != hipSuccess || hipStreamSynchronize((hipStream_t)blocked) != hipSuccess)
			return -1;
	}
	return 0;
}

int Launch::gate(const Params &p, void *stream, int publish_d, int need_d)
{
	hipLaunchKernelGGL(k_gate, dim3(1), dim3(64), 0, (hipStream_t)stream, p, publish_d, need_d);
	return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// k_intra_decide: open-loop intra mode decision from SOURCE samples, one wave
// per macroblock, every macroblock independent (DESIGN.md "decision / recon
// split"). Tile = source MB plus its top row, left column, top-left and four
// top-right samples.
// ------------------------------------------------------------------------
#define TS 24 // tile stride; tile[ty][tx] = sample (x0 - 1 + tx, y0 - 1 + ty)

__global__ __launch_bounds__(64) void k_intra_decide(Params p)
{
	__shared__ uint8_t s_t[17 * TS];       // luma tile with borders
	__shared__ uint8_t s_c[2][9 * 12];     // chroma tiles with borders, stride 12
	__shared__ uint8_t s_E[16][16];        // edge arrays per 4x4 block
	__shared__ int s_dcv[16];              // DC predictor value per block
	__shared__ uint32_t s_key[16][9];      // sad<<4 | mode per (block, mode)

	const int lane = threadIdx.x;
	int mx, my, st;
	xcd_mb(p, mx, my, st);
	const int x0 = mx * 16, y0 = my * 16;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, slice_of_row(p.mb_h, p.slice_count, my), &r0, &r1);
	const bool top_mb = my > r0, left_mb = mx > 0, tr_mb = top_mb && mx < p.mb_w - 1;
	const uint8_t *src_y = p.src_y + (size_t)st * p.plane_y_stride;
	const uint8_t *src_uv = p.src_uv + (size_t)st * p.plane_uv_stride;
	const size_t mbaddr = (size_t)st * p.n_mb + (size_t)my * p.mb_w + mx;

	// load tiles (byte loads; out-of-picture entries are never used)
	for (int i = lane; i < 17 * 21; i += 64) {
		int ty = i / 21, tx = i - ty * 21;
		int gx = x0 - 1 + tx, gy = y0 - 1 + ty;
		uint8_t v = 128;
		if (gx >= 0 && gy >= 0 && gx < p.cw && gy < p.ch)
			v = src_y[(size_t)gy * p.cw + gx];
		s_t[ty * TS + tx] = v;
	}
	for (int i = lane; i < 2 * 81; i += 64) {
		int c = i / 81, j = i - c * 81, ty = j / 9, tx = j - ty * 9;
		int gx = mx * 8 - 1 + tx, gy = my * 8 - 1 + ty;
		uint8_t v = 128;
		if (gx >= 0 && gy >= 0 && gx < (p.cw >> 1) && gy < (p.ch >> 1))
			v = src_uv[(size_t)gy * p.cw + 2 * gx + c];
		s_c[c][ty * 12 + tx] = v;
	}
	__syncthreads();

	// edge arrays + DC values, one lane per 4x4 block (coding order)
	if (lane < 16) {
		bool ht, hl, htr;
		blk4_avail(lane, top_mb, left_mb, tr_mb, ht, hl, htr);
		const int tx = 1 + 4 * blk_x(lane), ty = 1 + 4 * blk_y(lane);
		uint8_t *E = s_E[lane];
		int st_ = 0, sl = 0;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			uint8_t l = s_t[(ty + k) * TS + tx - 1], t = s_t[(ty - 1) * TS + tx + k];
			E[4 - k] = l;
			E[6 + k] = t;
			sl += l;
			st_ += t;
		}
		E[0] = E[1];
		E[5] = s_t[(ty - 1) * TS + tx - 1];
#pragma unroll
		for (int k = 4; k < 8; k++)
			E[6 + k] = htr ? s_t[(ty - 1) * TS + tx + k] : E[9];
		E[14] = E[13];
		s_dcv[lane] = (ht && hl) ? (st_ + sl + 4) >> 3 : hl ? (sl + 2) >> 2 : ht ? (st_ + 2) >> 2 : 128;
	}
	__syncthreads();

	// I4x4: 144 (mode, block) tasks over three rounds
	for (int round = 0; round < 3; round++) {
		const int t = round * 64 + lane;
		if (t < 144) {
			const int mode = t >> 4, b = t & 15;
			bool ht, hl, htr;
			blk4_avail(b, top_mb, left_mb, tr_mb, ht, hl, htr);
			uint32_t key = 0xffffffffu;
			if (mode4_allowed(mode, ht, hl)) {
				const int tx = 1 + 4 * blk_x(b), ty = 1 + 4 * blk_y(b);
				const uint8_t *E = s_E[b];
				const int dc = s_dcv[b];
				int sad = 0;
#pragma unroll
				for (int i = 0; i < 16; i++) {
					int pv = pred4_px(p.pred4_tab[mode * 16 + i], E, dc);
					sad += iabs((int)s_t[(ty + (i >> 2)) * TS + tx + (i & 3)] - pv);
				}
				key = ((uint32_t)sad << 4) | (uint32_t)mode;
			}
			s_key[b][mode] = key;
		}
	}
	__syncthreads();
	int sum4 = 0, best4 = 0;
	if (lane < 16) {
		uint32_t k = 0xffffffffu;
#pragma unroll
		for (int m = 0; m < 9; m++)
			k = s_key[lane][m] < k ? s_key[lane][m] : k;
		best4 = (int)(k & 15u);
		sum4 = (int)(k >> 4);
	}
	sum4 = wave_sum_i32(sum4);

	// I16x16: each lane covers row lane>>2, columns 4*(lane&3)..+3
	uint32_t best16 = 0xffffffffu;
	{
		const int py = lane >> 2, pxb = 4 * (lane & 3);
		int sadv = 0, sadh = 0, sadd = 0, sadp = 0;
		int dcsum = 0;
		// DC
		for (int k = 0; k < 16; k++)
			dcsum += (top_mb ? s_t[1 + k] : 0) + (left_mb ? s_t[(1 + k) * TS] : 0);
		const int dcv = (top_mb && left_mb) ? (dcsum + 16) >> 5 : (top_mb || left_mb) ? (dcsum + 8) >> 4 : 128;
		// plane parameters
		int H = 0, V = 0;
		for (int k = 0; k < 8; k++) {
			H += (k + 1) * ((int)s_t[1 + 8 + k] - (int)s_t[1 + 6 - k]);
			V += (k + 1) * ((int)s_t[(1 + 8 + k) * TS] - (int)s_t[(1 + 6 - k) * TS]);
		}
		const int a = 16 * ((int)s_t[16 * TS] + (int)s_t[16]);
		const int b = (5 * H + 32) >> 6, cc = (5 * V + 32) >> 6;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const int px = pxb + k;
			const int s = s_t[(1 + py) * TS + 1 + px];
			sadv += iabs(s - (int)s_t[1 + px]);
			sadh += iabs(s - (int)s_t[(1 + py) * TS]);
			sadd += iabs(s - dcv);
			sadp += iabs(s - clip255((a + b * (px - 7) + cc * (py - 7) + 16) >> 5));
		}
		sadv = wave_sum_i32(sadv);
		sadh = wave_sum_i32(sadh);
		sadd = wave_sum_i32(sadd);
		sadp = wave_sum_i32(sadp);
		if (top_mb)
			best16 = ((uint32_t)sadv << 2) | 0u;
		if (left_mb && (((uint32_t)sadh << 2) | 1u) < best16)
			best16 = ((uint32_t)sadh << 2) | 1u;
		if ((((uint32_t)sadd << 2) | 2u) < best16)
			best16 = ((uint32_t)sadd << 2) | 2u;
		if (top_mb && left_mb && (((uint32_t)sadp << 2) | 3u) < best16)
			best16 = ((uint32_t)sadp << 2) | 3u;
	}

	// chroma: lane = sample (x = lane&7, y = lane>>3), both planes
	uint32_t bestc = 0xffffffffu;
	{
		const int px = lane & 7, py = lane >> 3;
		int sad[4] = { 0, 0, 0, 0 };
#pragma unroll
		for (int c = 0; c < 2; c++) {
			const uint8_t *T = s_c[c];
			const int s = T[(1 + py) * 12 + 1 + px];
			// DC per 4x4 sub-block
			const int bx = px >> 2, by = py >> 2;
			int stp = 0, slf = 0;
			for (int k = 0; k < 4; k++) {
				stp += T[1 + 4 * bx + k];
				slf += T[(1 + 4 * by + k) * 12];
			}
			bool use_t = top_mb, use_l = left_mb;
			if (bx == 1 && by == 0 && top_mb)
				use_l = false;
			if (bx == 0 && by == 1 && left_mb)
				use_t = false;
			const int dcv = (use_t && use_l) ? (stp + slf + 4) >> 3 : use_t ? (stp + 2) >> 2 : use_l ? (slf + 2) >> 2 : 128;
			int H = 0, V = 0;
			for (int k = 0; k < 4; k++) {
				H += (k + 1) * ((int)T[1 + 4 + k] - (int)T[1 + 2 - k]);
				V += (k + 1) * ((int)T[(1 + 4 + k) * 12] - (int)T[(1 + 2 - k) * 12]);
			}
			const int a = 16 * ((int)T[8 * 12] + (int)T[8]);
			const int b = (34 * H + 32) >> 6, cc = (34 * V + 32) >> 6;
			sad[0] += iabs(s - dcv);
			sad[1] += iabs(s - (int)T[(1 + py) * 12]);
			sad[2] += iabs(s - (int)T[1 + px]);
			sad[3] += iabs(s - clip255((a + b * (px - 3) + cc * (py - 3) + 16) >> 5));
		}
#pragma unroll
		for (int m = 0; m < 4; m++)
			sad[m] = wave_sum_i32(sad[m]);
		bestc = ((uint32_t)sad[0] << 2) | 0u;
		if (left_mb && (((uint32_t)sad[1] << 2) | 1u) < bestc)
			bestc = ((uint32_t)sad[1] << 2) | 1u;
		if (top_mb && (((uint32_t)sad[2] << 2) | 2u) < bestc)
			bestc = ((uint32_t)sad[2] << 2) | 2u;
		if (top_mb && left_mb && (((uint32_t)sad[3] << 2) | 3u) < bestc)
			bestc = ((uint32_t)sad[3] << 2) | 3u;
	}

	if (lane < 16)
		p.i4modes[mbaddr * 16 + lane] = (int8_t)best4;
	if (lane == 0) {
		MbInfo m;
		m.type = (sum4 + 32 * p.lambda < (int)(best16 >> 2)) ? MB_I4x4 : MB_I16x16;
		m.i16_mode = (int8_t)(best16 & 3u);
		m.chroma_mode = (int8_t)(bestc & 3u);
		m.cbp = 0;
		m.mvx = 0;
		m.mvy = 0;
		p.mbi[mbaddr] = m;
	}
}

int Launch::intra_decide(const Params &p, void *stream)
{
	dim3 grid(p.mb_w * p.row_num * p.n_streams);
	hipLaunchKernelGGL(k_intra_decide, grid, dim3(64), 0, (hipStream_t)stream, p);
	return (int)hipGetLastError();
}

} // namespace cedar
