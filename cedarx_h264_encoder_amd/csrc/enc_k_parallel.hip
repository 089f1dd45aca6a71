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
#include "enc_cavlc_len.h"

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
	if (blockIdx.y == 0 && st == 0 && seg == 0 && p.cut_votes) // scene cuts: this frame's count starts at zero
		p.cut_votes[0] = 0;
	if (blockIdx.x == 0 && blockIdx.y == 0 && st == 0 && threadIdx.x == 0)
		dbg_event(p, 1, p.db_tag, 0);
	if (blockIdx.y == 0 && st == 0 && (p.wf_split_intra > 1 || p.ir_rows)) {
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
// per-wave LDS arena of k_inter (bytes)
#define S8_STRIDE 96                       // bytes per row of the 8x8 box sums: 48 x u16
#define IA_REF 0                           // 48 x 48 window + 64 bytes of slack
#define IA_CUR (48 * REF_STRIDE + 64)      // 16 x 16 source samples
#define IA_CURC (IA_CUR + 256)             // 8 rows x 8 CbCr pairs
#define IA_U (IA_CURC + 128)               // the union: box sums (41 rows), then the survivor list and the cost tables ...
#define IA_LIST 0                          //   u16[320]
#define IA_LAMX 640                        //   u32[36]
#define IA_LAMY (640 + 144)                //   u32[36]
#define IA_REFC 0                          // ... then the chroma window, the levels, the reconstruction, the counts
#define IA_COEF (25 * REFC_STRIDE)
#define IA_RECY (IA_COEF + 816)
#define IA_RECC (IA_RECY + 256)
#define IA_NNZ (IA_RECC + 128)
#define IA_BYTES (IA_U + 41 * S8_STRIDE)
static_assert(IA_NNZ + 32 <= 41 * S8_STRIDE && (IA_U & 15) == 0 && (IA_COEF & 15) == 0 && (IA_BYTES & 15) == 0, "k_inter arena");

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
	const int xs = p.stripe_s0 - my; // stripes of x + y
	mx = (xs > 0 ? xs : 0) + (int)(m - p.stripe_tab[lo]);
	return st < p.n_streams;
}

// LDS traffic of one wave is private to it here: order it with a wave-level fence instead of the
// workgroup barrier (which would make the four macroblocks of a workgroup wait for each other)
#define WAVE_LDS_SYNC()                                                        \
	do {                                                                   \
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local"); \
		__builtin_amdgcn_wave_barrier();                               \
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront", "local"); \
	} while (0)

typedef uint64_t u64_a4 __attribute__((aligned(4)));

// raster position of a coefficient -> its zig-zag scan index, one nibble each
DEVI int zigzag_of(int rpos) { return (int)((0xFEA9DB83C7426510ull >> (4 * rpos)) & 15u); }

// transforms of enc_device.h with the lane-constant multiplies as 24-bit multiply-adds (one
// full-rate instruction each; v_mul_lo_u32 is quarter rate): every operand is far inside 24 bits
// (|residual| <= 255, |W| <= 9180, |scaled coefficient| < 2^17)
DEVI int fdct_lanes24(int v, const LaneK &ci, const LaneK &cj)
{
	int s = __mul24(cj.s1, v) + dpp_xor3(v);
	int q = dpp_xor1(s);
	s = __mul24(cj.fa, s) + __mul24(cj.fb, q);
	s = __mul24(ci.s1, s) + dpp_xor12(s);
	q = dpp_xor4(s);
	return __mul24(ci.fa, s) + __mul24(ci.fb, q);
}

DEVI int idct_lanes24(int d, const LaneK &ci, const LaneK &cj)
{
	int q = dpp_xor1(d);
	int e = __mul24(cj.ia, d >> cj.sh) + __mul24(cj.ib, q);
	int f = __mul24(cj.s1, e) + dpp_xor3(e);
	q = dpp_xor4(f);
	e = __mul24(ci.ia, f >> ci.sh) + __mul24(ci.ib, q);
	f = __mul24(ci.s1, e) + dpp_xor12(e);
	return (f + 32) >> 6;
}

// quant1 of enc_device.h for |w| < 2^15, mf < 2^14: the product fits the 24-bit multiplier
DEVI int quant24(int w, int mf, int f, int qbits)
{
	const uint32_t a = (uint32_t)iabs(w);
	int l = (int)((__umul24(a, (uint32_t)mf) + (uint32_t)f) >> qbits);
	l = l > LEVEL_MAX ? LEVEL_MAX : l;
	return w < 0 ? -l : l;
}

// ---- sub-sample refinement (Params::subpel; clause 8.4.2.2.1) ----
// Around the best whole-sample vector the half-sample planes of the reference are built once in LDS:
//   b  half a sample to the right of G      h  half a sample below G      j  both
// (b1 = the unrounded 6-tap row sums that j is filtered from). Every fractional prediction is one of
// G / b / h / j or the rounded mean of two of them (the standard's a..s table), so a candidate is
// two (plane, +0/+1 column, +0/+1 row) picks, kSubTab below; its SAD is one v_sad_u8 per lane
// (16 rows x 4 segments of 4 samples = 64 lanes) and a wave sum.
#define SP_STRIDE 20 // bytes per row of the b / h / j planes; plane sample (u, v), u, v >= -1, at [(v + 1) * 20 + u + 1]
#define SP_B1_STRIDE 18 // int16 per row of b1; (u, v), u >= -1, v >= -3, at [(v + 3) * 18 + u + 1]
struct SubPick { int8_t pa, ua, va, pb, ub, vb; }; // planes: 0 G, 1 b, 2 h, 3 j
// index = 4 * fy + fx
__device__ constexpr SubPick kSubTab[16] = {
	{ 0, 0, 0, 0, 0, 0 }, { 0, 0, 0, 1, 0, 0 }, { 1, 0, 0, 1, 0, 0 }, { 0, 1, 0, 1, 0, 0 }, // G   a   b   c
	{ 0, 0, 0, 2, 0, 0 }, { 1, 0, 0, 2, 0, 0 }, { 1, 0, 0, 3, 0, 0 }, { 1, 0, 0, 2, 1, 0 }, // d   e   f   g
	{ 2, 0, 0, 2, 0, 0 }, { 2, 0, 0, 3, 0, 0 }, { 3, 0, 0, 3, 0, 0 }, { 2, 1, 0, 3, 0, 0 }, // h   i   j   k
	{ 0, 0, 1, 2, 0, 0 }, { 1, 0, 1, 2, 0, 0 }, { 1, 0, 1, 3, 0, 0 }, { 1, 0, 1, 2, 1, 0 }, // n   p   q   r
};

// bits of se(v), v a vector component in quarter samples (mv_bits(d) for v = 4 d)
DEVI int mv_bits_q(int v)
{
	const int a = iabs(v);
	return a == 0 ? 1 : 2 * (31 - __clz(2 * a)) + 1;
}

// four bytes from a byte address in LDS that need not be aligned: two aligned dwords and a funnel shift
DEVI uint32_t lds_bytes4(const uint8_t *base, int off)
{
	const uint32_t *w = (const uint32_t *)(base + (off & ~3));
	return __builtin_amdgcn_alignbyte(w[1], w[0], (uint32_t)off & 3u);
}

// rounded mean of four byte pairs: (a + b + 1) >> 1 per byte
DEVI uint32_t avg_bytes4(uint32_t a, uint32_t b) { return (a | b) - (((a ^ b) >> 1) & 0x7f7f7f7fu); }

// What a macroblock's QP decides. Without regions of interest it is the picture's and sits in Params;
// with them the macroblock's QP_Y picks a row of the session's table (and the row of its QP_C).
struct QpSet {
	int lambda, qbits, qshift, qbits_c, qshift_c;
	int mf[3], dv[3], mfc[3], dvc[3];
};

template <bool MBQP>
DEVI QpSet qp_set(const Params &p, int qp_mb)
{
	QpSet q;
	if (MBQP) {
		const QpRow &y = p.qtab[qp_mb];
		const QpRow &c = p.qtab[y.qpc];
		q.lambda = y.lambda; q.qbits = y.qbits; q.qshift = y.qshift; q.qbits_c = c.qbits; q.qshift_c = c.qshift;
#pragma unroll
		for (int k = 0; k < 3; k++) {
			q.mf[k] = y.mf[k]; q.dv[k] = y.dv[k]; q.mfc[k] = c.mf[k]; q.dvc[k] = c.dv[k];
		}
	} else {
		q.lambda = p.lambda; q.qbits = p.qbits; q.qshift = p.qshift; q.qbits_c = p.qbits_c; q.qshift_c = p.qshift_c;
#pragma unroll
		for (int k = 0; k < 3; k++) {
			q.mf[k] = p.mf[k]; q.dv[k] = p.dv[k]; q.mfc[k] = p.mfc[k]; q.dvc[k] = p.dvc[k];
		}
	}
	return q;
}

// MBQP: macroblocks differ in QP (regions of interest, MAD classes) -- a kernel of its own, like the
// sub-sample refinement and the refresh branch, so that the plain one keeps its registers
// What k_gate (below) waits for, as one wave of another kernel: every deblock band of the previous picture that owns diagonals
// below need_d has written them back. No acquire here: the caller reads nothing after it -- the next launch on the stream
// starts behind it with the acquire of a kernel's start.
DEVI void wait_prev_deblock(const Params &p, int lane, int need_d)
{
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
			pending = band < ((r1 - r0 + 3) >> 2) && r0 + 4 * band < need_d;
		}
		uint32_t spins = 0;
		uint64_t t0 = 0;
		while (true) {
			if (pending) {
				const uint32_t v = __hip_atomic_load(&p.prev_db[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				pending = (int32_t)(v - (p.prev_tag | (uint32_t)need_d)) < 0;
			}
			if (!__any(pending))
				break;
			if (wait_expired(++spins, t0) || ((spins & 63u) == 0 && __hip_atomic_load(p.abort_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
				if (lane == 0) {
					__hip_atomic_store(p.wf_err, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					__hip_atomic_store(p.abort_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				return;
			}
			__builtin_amdgcn_s_sleep(32);
		}
	}
}

template <int SUBPEL, bool REFRESH, bool MBQP>
__global__ __launch_bounds__(64 * INTER_WAVES) void k_inter(Params p)
{
	// one wave per macroblock, INTER_WAVES macroblocks per workgroup; each wave has its own
	// arena and never looks at another wave's. The box sums of the search and what the transform
	// stage needs never live at the same time: one union (6 workgroups fit the CU's 160 KB).
	__shared__ __attribute__((aligned(16))) uint8_t sa_arena[INTER_WAVES][IA_BYTES];
	// half-sample planes b, h, j (18 rows each, 4 bytes of slack for the funnel-shift reads)
	__shared__ __attribute__((aligned(16))) uint8_t sa_sub[SUBPEL ? INTER_WAVES : 1][SUBPEL ? 3 * 18 * SP_STRIDE + 16 : 16];

	const int lane = threadIdx.x & 63;
	const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform, and the compiler knows
	if (blockIdx.x == 0 && threadIdx.x == 0)
		dbg_event(p, 9, p.db_tag, (uint32_t)p.stripe_s0);
	// (tail_gate: the launches before this one have completed -- stream order --, the stripe before this one among them)
	if (blockIdx.x == 0 && threadIdx.x == 0 && p.gate_publish >= 0 && p.prog_inter)
		__hip_atomic_store(p.prog_inter, p.prog_tag | (uint32_t)p.gate_publish, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
	uint8_t *const arena = sa_arena[wave];
	uint8_t *const s_ref = arena + IA_REF, *const s_cur = arena + IA_CUR, *const s_curc = arena + IA_CURC;
	uint8_t *const s_s8 = arena + IA_U;                              // search only
	uint8_t *const s_refc = arena + IA_U + IA_REFC;                  // from here on: after the search
	int16_t *const s_coef = (int16_t *)(arena + IA_U + IA_COEF);
	uint8_t *const s_recy = arena + IA_U + IA_RECY, *const s_recc = arena + IA_U + IA_RECC;
	uint8_t *const s_nnz = arena + IA_U + IA_NNZ;
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
	// Cyclic intra refresh (Params::intra_refresh = N, refresh_t = P pictures since the IDR): this
	// macroblock is coded Intra16x16 (DC) when (x + 2*y + t) % N == 0. Its neighbours are inter, hence
	// "not available" under constrained_intra_pred: the prediction is 128 everywhere, luma and chroma.
	const bool refresh = REFRESH && (mx + 2 * my + p.refresh_t) % p.intra_refresh == 0; // wave-uniform
	// everything that follows from the QP: the picture's, or this macroblock's (regions of interest)
	const int qp_base = MBQP ? roi_qp(p, st, mx, my) : p.qp;
	QpSet qs = qp_set<MBQP>(p, qp_base);

	const uint8_t *src_y = src_luma(p, st);
	const uint8_t *src_uv = src_chroma(p, st);
	const uint8_t *ref_y = p.ref_y + (size_t)st * p.plane_y_stride;
	const uint8_t *ref_uv = p.ref_uv + (size_t)st * p.plane_uv_stride;
	uint8_t *rec_y = valid ? p.rec_y + (size_t)st * p.plane_y_stride + (size_t)y0 * p.cw + x0 : p.dummy_rec;
	uint8_t *rec_uv = valid ? p.rec_uv + (size_t)st * p.plane_uv_stride + (size_t)(my * 8) * p.cw + x0 : p.dummy_rec + (size_t)16 * p.cw;
	// records, counts and levels of the dummy macroblock live one past the last stream
	const size_t mbaddr = valid ? (size_t)st * p.n_mb + (size_t)my * p.mb_w + mx : (size_t)p.dummy_mb;

	// ---- stage tiles through LDS with 16-byte loads (offsets inside a plane fit 32 bits) ----
	const uint32_t cw = (uint32_t)p.cw;
	if (lane < 16)
		*(u32x4 *)(s_cur + lane * 16) = *(const u32x4 *)(src_y + (uint32_t)min(y0 + lane, p.src_last_y) * cw + (uint32_t)x0);
	else if (lane < 24)
		*(u32x4 *)(s_curc + (lane - 16) * 16) = *(const u32x4 *)(src_uv + (uint32_t)min(my * 8 + lane - 16, p.src_last_c) * cw + (uint32_t)x0);
	// Reference window. Samples outside the picture are never used (a candidate that touches them
	// is not valid), so the loads are unconditional from clamped coordinates: no branches.
	// chroma: 25 rows x 4 segments; lane -> (row = lane / 4 + 16 * pass, seg = lane % 4). It waits in
	// registers until the search is over (its place in LDS holds the search's box sums until then).
	const int crow = lane >> 2, cseg = lane & 3;
	u32x4 cwin[2];
	{
		// luma: 48 rows x 3 segments of 16 bytes; lane -> (row = lane / 3 + 21 * pass, seg = lane % 3), 63 lanes
		const int wrow = min(lane, 62) / 3, wseg = min(lane, 62) - 3 * wrow;
		const uint32_t gx = (uint32_t)min(max(x0 - 16 + 16 * wseg, 0), p.cw - 16);
		u32x4 v[3];
#pragma unroll
		for (int pass = 0; pass < 3; pass++) {
			const int gy = min(max(y0 - 16 + wrow + 21 * pass, 0), p.ch - 1);
			v[pass] = *(const u32x4 *)(ref_y + (uint32_t)gy * cw + gx);
		}
		const uint32_t cgx = (uint32_t)min(max(x0 - 16 + 16 * cseg, 0), p.cw - 16);
#pragma unroll
		for (int pass = 0; pass < 2; pass++) {
			const int gy = min(max(my * 8 - 8 + crow + 16 * pass, 0), (p.ch >> 1) - 1);
			cwin[pass] = *(const u32x4 *)(ref_uv + (uint32_t)gy * cw + cgx);
		}
		*(u32x4 *)(s_ref + wrow * REF_STRIDE + 16 * wseg) = v[0];
		*(u32x4 *)(s_ref + (wrow + 21) * REF_STRIDE + 16 * wseg) = v[1];
		if (wrow < 6) // rows 42..47 (lane 63 repeats lane 62's store)
			*(u32x4 *)(s_ref + (wrow + 42) * REF_STRIDE + 16 * wseg) = v[2];
		// (the last dx group reads one dword past row 47: the 64 bytes of slack behind the window)
	}
	WAVE_LDS_SYNC();

	// ---- motion search: exhaustive in its RESULT, successive elimination in its work ----
	// For any candidate, sum over the four 8x8 quadrants |sum(cur quadrant) - sum(ref quadrant)| <= SAD
	// (triangle inequality), so "bound + lambda * bits > the cost of some candidate already evaluated"
	// proves that the candidate cannot be the minimum, nor tie with it. The bounds of all 33 x 33
	// candidates cost about a tenth of their SADs: (A) 8x8 box sums of the window, once; (B) per
	// candidate 2 v_sad_u16. (C) The quad of 4 horizontal neighbours with the least bound is evaluated
	// exactly, its best cost is the threshold; (D) every quad holding a candidate whose bound does not
	// exceed it is evaluated exactly (v_qsad_pk_u16_u8: 4 neighbours per instruction), 16 quads per pass, as many passes as it takes:
	// noise keeps all 297 quads and costs about what summing everything did. The minimum of
	// cost << 11 | index over the evaluated candidates IS the exhaustive search's (ties to the lowest
	// index): nothing that could win or tie is ever dropped.
	uint32_t cs0, cs1; // 8x8 sums of the current macroblock: {top left, top right}, {bottom left, bottom right} as u16 pairs
	{
		// lane = (quadrant b = lane / 16, dword r of the quadrant): row 8 * (b / 2) + r / 2, dwords 2 * (b % 2) + r % 2
		const int b = lane >> 4, r = lane & 15;
		const uint32_t mine_q = ((const uint32_t *)s_cur)[(8 * (b >> 1) + (r >> 1)) * 4 + 2 * (b & 1) + (r & 1)];
		int q = (int)__builtin_amdgcn_sad_u8(mine_q, 0u, 0u);
		q += dpp_xor1(q);
		q += __builtin_amdgcn_update_dpp(0, q, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
		q += __builtin_amdgcn_update_dpp(0, q, 0x141, 0xf, 0xf, true); // row_half_mirror
		q += __builtin_amdgcn_update_dpp(0, q, 0x140, 0xf, 0xf, true); // row_mirror
		cs0 = (uint32_t)__builtin_amdgcn_readlane(q, 0) | (uint32_t)__builtin_amdgcn_readlane(q, 16) << 16;
		cs1 = (uint32_t)__builtin_amdgcn_readlane(q, 32) | (uint32_t)__builtin_amdgcn_readlane(q, 48) << 16;
	}
	{
		// (A) S8[y][x] = sum of the 8x8 block of the window at (x, y), y in [0, 41), x in [0, 48) (x > 40
		// holds junk that only candidates beyond the range read), u16. A lane owns 4 neighbouring x (g)
		// and 11 rows (seg): per window row one v_qsad against zero gives the four 4-sample sums, the
		// lane to the right has the next four (DPP), a running sum of 8 rows does the rest. u16 pairs
		// are added as dwords: no sum reaches 2^16.
		const int seg = lane >> 4, g = min(lane & 15, 11);
		const uint8_t *hp = s_ref + (10 * seg) * REF_STRIDE + 4 * g;
		uint8_t *wp = s_s8 + (10 * seg) * S8_STRIDE + 8 * g;
		uint32_t hlo[8], hhi[8], slo = 0, shi = 0;
#pragma unroll
		for (int t = 0; t < 18; t++) {
			const uint64_t w = *(const u64_a4 *)(hp + t * REF_STRIDE);
			const uint64_t h4 = __builtin_amdgcn_qsad_pk_u16_u8(w, 0u, 0ull);
			uint32_t lo = (uint32_t)h4, hi = (uint32_t)(h4 >> 32);
			lo += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)lo, 0x101, 0xf, 0xf, true); // row_shl:1: the lane to the right
			hi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)hi, 0x101, 0xf, 0xf, true);
			if (t >= 8) {
				slo -= hlo[t & 7];
				shi -= hhi[t & 7];
			}
			slo += lo;
			shi += hi;
			hlo[t & 7] = lo;
			hhi[t & 7] = hi;
			if (t >= 7)
				*(uint64_t *)(wp + (t - 7) * S8_STRIDE) = (uint64_t)slo | (uint64_t)shi << 32;
		}
	}
	WAVE_LDS_SYNC();
	const int R = p.search_range;
	const int g9 = min(lane, 62) % 9, j7 = min(lane, 62) / 9;
	const int d0 = min(5 * j7, 28);
	uint32_t bnd[5][4];
	{
		// (B) a lane owns the quads dx = 4 * g9 - 16 .. + 3 at dy = d0 - 16 .. + 4 (as the exhaustive loop
		// did). Row Y of the sums serves candidate row Y (upper quadrants) and Y - 8 (lower ones).
		const uint8_t *bp = s_s8 + d0 * S8_STRIDE + 8 * g9;
#pragma unroll
		for (int t = 0; t < 13; t++) {
			if (t >= 5 && t < 8)
				continue;
			const uint64_t a = *(const uint64_t *)(bp + t * S8_STRIDE), b = *(const uint64_t *)(bp + t * S8_STRIDE + 16);
			const uint32_t a0 = (uint32_t)a, a1 = (uint32_t)(a >> 32), b0 = (uint32_t)b, b1 = (uint32_t)(b >> 32);
			uint32_t pr[4];
			pr[0] = __builtin_amdgcn_perm(b0, a0, 0x05040100u); // {S8[Y][X], S8[Y][X + 8]}
			pr[1] = __builtin_amdgcn_perm(b0, a0, 0x07060302u);
			pr[2] = __builtin_amdgcn_perm(b1, a1, 0x05040100u);
			pr[3] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
#pragma unroll
			for (int k = 0; k < 4; k++)
				bnd[t < 5 ? t : t - 8][k] = t < 5 ? __builtin_amdgcn_sad_u16(pr[k], cs0, 0u) : __builtin_amdgcn_sad_u16(pr[k], cs1, bnd[t - 8][k]);
		}
	}
	WAVE_LDS_SYNC(); // the sums are dead: their place takes the cost tables and the list
	uint32_t *const s_lamx = (uint32_t *)(s_s8 + IA_LAMX), *const s_lamy = (uint32_t *)(s_s8 + IA_LAMY);
	uint16_t *const s_list = (uint16_t *)(s_s8 + IA_LIST);
	{
		// lambda * bits of a vector component; a component that is not allowed costs 2^19 (any valid
		// cost stays below 2^17; the sum of both still fits cost << 11)
		const int d = lane - 16;
		const bool xok = d >= -R && d <= R && x0 + d >= 0 && x0 + d + 16 <= p.cw;
		const bool yok = d >= -R && d <= R && y0 + d >= band_y0 && y0 + d + 16 <= band_y1;
		const uint32_t lb = (uint32_t)(qs.lambda * mv_bits(d));
		if (lane < 36) {
			s_lamx[lane] = lb + (xok ? 0u : 0x80000u);
			s_lamy[lane] = lb + (yok ? 0u : 0x80000u);
		}
	}
	WAVE_LDS_SYNC();
	uint32_t best;
	int n_list;
	{
		// (C) bound cost of a quad = min over its four candidates; key = cost << 9 | quad (row * 9 + g)
		const u32x4 lx = *(const u32x4 *)(s_lamx + 4 * g9);
		uint32_t qb[5], kmin = 0xffffffffu;
#pragma unroll
		for (int i = 0; i < 5; i++) {
			const uint32_t m = min(min(bnd[i][0] + lx.x, bnd[i][1] + lx.y), min(bnd[i][2] + lx.z, bnd[i][3] + lx.w));
			qb[i] = m + s_lamy[d0 + i];
			kmin = min(kmin, (qb[i] << 9) + (uint32_t)((d0 + i) * 9 + g9));
		}
		kmin = wave_min_u32(kmin);
		const int q0 = (int)(kmin & 511u), qy0 = q0 / 9, qg0 = q0 - 9 * qy0; // wave-uniform
		// that quad exactly: 16 lanes = its 16 rows, summed across the lanes of a DPP row
		uint32_t thr_key;
		{
			const int r = lane & 15;
			const uint8_t *rp = s_ref + (qy0 + r) * REF_STRIDE + 4 * qg0;
			const u32x4 cv = *(const u32x4 *)(s_cur + 16 * r);
			uint64_t a = __builtin_amdgcn_qsad_pk_u16_u8(*(const u64_a4 *)(rp), cv.x, 0ull);
			a = __builtin_amdgcn_qsad_pk_u16_u8(*(const u64_a4 *)(rp + 4), cv.y, a);
			a = __builtin_amdgcn_qsad_pk_u16_u8(*(const u64_a4 *)(rp + 8), cv.z, a);
			a = __builtin_amdgcn_qsad_pk_u16_u8(*(const u64_a4 *)(rp + 12), cv.w, a);
			int lo = (int)(uint32_t)a, hi = (int)(uint32_t)(a >> 32); // u16 pairs: a 16x16 SAD is at most 65280, no carry
			lo += dpp_xor1(lo); hi += dpp_xor1(hi);
			lo += __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, true); hi += __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, true);
			lo += __builtin_amdgcn_update_dpp(0, lo, 0x141, 0xf, 0xf, true); hi += __builtin_amdgcn_update_dpp(0, hi, 0x141, 0xf, 0xf, true);
			lo += __builtin_amdgcn_update_dpp(0, lo, 0x140, 0xf, 0xf, true); hi += __builtin_amdgcn_update_dpp(0, hi, 0x140, 0xf, 0xf, true);
			const u32x4 l0 = *(const u32x4 *)(s_lamx + 4 * qg0);
			const uint32_t ly = s_lamy[qy0], ib = (uint32_t)(qy0 * 33 + 4 * qg0);
			thr_key = min(min(((((uint32_t)lo & 0xffffu) + l0.x + ly) << 11) + ib, ((((uint32_t)lo >> 16) + l0.y + ly) << 11) + ib + 1u),
				      min(((((uint32_t)hi & 0xffffu) + l0.z + ly) << 11) + ib + 2u, ((((uint32_t)hi >> 16) + l0.w + ly) << 11) + ib + 3u));
			thr_key = (uint32_t)__builtin_amdgcn_readfirstlane((int)thr_key);
		}
		best = thr_key;
		const uint32_t thr = thr_key >> 11;
		// survivors -> list (lane 63 mirrors lane 62 and the last range of rows overlaps the one before
		// it by two: neither enters twice; the quad already done stays out)
		int base = 0;
#pragma unroll
		for (int i = 0; i < 5; i++) {
			const bool f = qb[i] <= thr && lane < 63 && !(j7 == 6 && i < 2) && (d0 + i) * 9 + g9 != q0;
			const uint64_t m = __ballot(f);
			const int pos = base + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
			if (f)
				s_list[pos] = (uint16_t)(((d0 + i) << 4) | g9);
			base += __popcll(m);
		}
		n_list = base;
	}
	WAVE_LDS_SYNC();
	{
		// (D) four lanes per quad (4 rows each: 16 v_qsad), 16 quads per pass, as many passes as it takes
		// (slots past the end redo the last quad)
		const int slot = lane >> 2, part = lane & 3;
		u32x4 cv[4];
#pragma unroll
		for (int r = 0; r < 4; r++)
			cv[r] = *(const u32x4 *)(s_cur + 16 * (4 * part + r));
		for (int base = 0; base < n_list; base += 16) {
			const int e = (int)s_list[min(base + slot, n_list - 1)];
			const int qy = e >> 4, qg = e & 15;
			const uint8_t *rp = s_ref + (qy + 4 * part) * REF_STRIDE + 4 * qg;
			uint64_t a0 = 0, a1 = 0;
#pragma unroll
			for (int r = 0; r < 4; r++) {
				a0 = __builtin_amdgcn_qsad_pk_u16_u8(*(const u64_a4 *)(rp + r * REF_STRIDE), cv[r].x, a0);
				a1 = __builtin_amdgcn_qsad_pk_u16_u8(*(const u64_a4 *)(rp + r * REF_STRIDE + 4), cv[r].y, a1);
				a0 = __builtin_amdgcn_qsad_pk_u16_u8(*(const u64_a4 *)(rp + r * REF_STRIDE + 8), cv[r].z, a0);
				a1 = __builtin_amdgcn_qsad_pk_u16_u8(*(const u64_a4 *)(rp + r * REF_STRIDE + 12), cv[r].w, a1);
			}
			// per-u16 sums cannot carry: a full 16x16 SAD is at most 65280
			int lo = (int)((uint32_t)a0 + (uint32_t)a1), hi = (int)((uint32_t)(a0 >> 32) + (uint32_t)(a1 >> 32));
			lo += dpp_xor1(lo); hi += dpp_xor1(hi);
			lo += __builtin_amdgcn_update_dpp(0, lo, 0x4E, 0xf, 0xf, true); hi += __builtin_amdgcn_update_dpp(0, hi, 0x4E, 0xf, 0xf, true);
			const u32x4 lx = *(const u32x4 *)(s_lamx + 4 * qg);
			const uint32_t ly = s_lamy[qy], ib = (uint32_t)(qy * 33 + 4 * qg);
			best = min(best, min(min(((((uint32_t)lo & 0xffffu) + lx.x + ly) << 11) + ib, ((((uint32_t)lo >> 16) + lx.y + ly) << 11) + ib + 1u),
					     min(((((uint32_t)hi & 0xffffu) + lx.z + ly) << 11) + ib + 2u, ((((uint32_t)hi >> 16) + lx.w + ly) << 11) + ib + 3u)));
		}
	}
	WAVE_LDS_SYNC(); // list and tables are dead: the chroma window takes their place
	*(u32x4 *)(s_refc + crow * REFC_STRIDE + 16 * cseg) = cwin[0];
	if (crow < 9) // rows 16..24
		*(u32x4 *)(s_refc + (crow + 16) * REFC_STRIDE + 16 * cseg) = cwin[1];
	best = wave_min_u32(best);
	const int bidx = (int)(best & 2047u);
	const int bdy = bidx / 33 - 16, bdx = bidx - (bidx / 33) * 33 - 16;
	int qx = (REFRESH && refresh) ? 0 : 4 * bdx, qy = (REFRESH && refresh) ? 0 : 4 * bdy; // the vector in quarter samples
	uint8_t *const s_sub = sa_sub[SUBPEL ? wave : 0];
	uint32_t final_cost = best >> 11; // SAD + lambda * bits of the vector that is kept
	if (SUBPEL && p.subpel > 0 && !(REFRESH && refresh)) {
		// window coordinates of the best whole-sample block; plane sample (u, v) = window (wx0 + u, wy0 + v)
		const int wx0 = 16 + bdx, wy0 = 16 + bdy;
		int16_t *const s_b1 = (int16_t *)s_coef; // 22 rows x 18: the levels' place is free until the transform
		uint8_t *const s_pb = s_sub, *const s_ph = s_sub + 18 * SP_STRIDE, *const s_pj = s_sub + 36 * SP_STRIDE;
		// b1 (u in [-1, 15], v in [-3, 18]) and b = clip((b1 + 16) >> 5) for v in [-1, 16]. Addresses are
		// clamped into the window: a clamped tap is a wrong tap, but only candidates that subpel_ok
		// refuses read it.
		for (int i = lane; i < 22 * 17; i += 64) {
			const int v = i / 17 - 3, u = i - (v + 3) * 17 - 1;
			const uint8_t *r = s_ref + min(max(wy0 + v, 0), 47) * REF_STRIDE + min(max(wx0 + u - 2, 0), 42);
			const int t = (int)r[0] - 5 * (int)r[1] + 20 * (int)r[2] + 20 * (int)r[3] - 5 * (int)r[4] + (int)r[5];
			s_b1[(v + 3) * SP_B1_STRIDE + u + 1] = (int16_t)t;
			if (v >= -1 && v <= 16)
				s_pb[(v + 1) * SP_STRIDE + u + 1] = (uint8_t)clip255((t + 16) >> 5);
		}
		// h (u in [-1, 16], v in [-1, 15])
		for (int i = lane; i < 17 * 18; i += 64) {
			const int v = i / 18 - 1, u = i - (v + 1) * 18 - 1;
			const uint8_t *r = s_ref + min(max(wy0 + v - 2, 0), 42) * REF_STRIDE + min(max(wx0 + u, 0), 47);
			const int t = (int)r[0] - 5 * (int)r[REF_STRIDE] + 20 * (int)r[2 * REF_STRIDE] + 20 * (int)r[3 * REF_STRIDE] -
				      5 * (int)r[4 * REF_STRIDE] + (int)r[5 * REF_STRIDE];
			s_ph[(v + 1) * SP_STRIDE + u + 1] = (uint8_t)clip255((t + 16) >> 5);
		}
		WAVE_LDS_SYNC();
		// j (u, v in [-1, 15]) from six b1 of a column
		for (int i = lane; i < 17 * 17; i += 64) {
			const int v = i / 17 - 1, u = i - (v + 1) * 17 - 1;
			const int16_t *c = s_b1 + (v + 1) * SP_B1_STRIDE + u + 1; // row v - 2
			const int t = (int)c[0] - 5 * (int)c[SP_B1_STRIDE] + 20 * (int)c[2 * SP_B1_STRIDE] + 20 * (int)c[3 * SP_B1_STRIDE] -
				      5 * (int)c[4 * SP_B1_STRIDE] + (int)c[5 * SP_B1_STRIDE];
			s_pj[(v + 1) * SP_STRIDE + u + 1] = (uint8_t)clip255((t + 512) >> 10);
		}
		WAVE_LDS_SYNC();
		// candidates: lane = (row r = lane / 4, segment c = lane % 4) of the block
		const int sr = lane >> 2, sc = lane & 3;
		const uint32_t cur4 = ((const uint32_t *)s_cur)[lane]; // row sr, samples 4 sc .. 4 sc + 3
		auto pred4 = [&](int vx, int vy) -> uint32_t { // the 4 predicted samples of this lane for vector (vx, vy)
			const int ox = (vx >> 2) - bdx, oy = (vy >> 2) - bdy; // -1 or 0
			const SubPick k = kSubTab[4 * (vy & 3) + (vx & 3)];
			auto fetch = [&](int pl, int du, int dv) -> uint32_t {
				const int u = 4 * sc + ox + du, v = sr + oy + dv;
				if (pl == 0) // (never below the window's first byte: a refused candidate may point there)
					return lds_bytes4(s_ref, max((wy0 + v) * REF_STRIDE + wx0 + u, 0));
				return lds_bytes4(s_sub, (pl - 1) * 18 * SP_STRIDE + (v + 1) * SP_STRIDE + u + 1);
			};
			return avg_bytes4(fetch(k.pa, k.ua, k.va), fetch(k.pb, k.ub, k.vb));
		};
		auto allowed = [&](int vx, int vy) -> bool { // see subpel_ok of the scalar twin
			const int ix = vx >> 2, iy = vy >> 2, fx = vx & 3, fy = vy & 3;
			const int X = x0 + ix, Y = y0 + iy;
			bool ok = vx >= -4 * R && vx <= 4 * R && vy >= -4 * R && vy <= 4 * R;
			ok = ok && X >= 0 && X + 16 <= p.cw && Y >= band_y0 && Y + 16 <= band_y1;
			ok = ok && (!fx || (X - 2 >= 0 && X + 18 < p.cw && ix >= -14 && ix <= 13));
			ok = ok && (!fy || (Y - 2 >= band_y0 && Y + 18 < band_y1 && iy >= -14 && iy <= 13));
			return ok;
		};
		uint32_t bestj = final_cost; // SAD + lambda * bits of the incumbent
		for (int stage = 1; stage <= p.subpel && stage <= 2; stage++) {
			const int step = stage == 1 ? 2 : 1, cx = qx, cy = qy;
			uint32_t key = bestj << 4; // incumbent: index 0, wins ties
#pragma unroll
			for (int pr = 0; pr < 4; pr++) { // two candidates per wave sum (a lane's SAD is < 2^16 / 64)
				uint32_t packed = 0;
#pragma unroll
				for (int hh = 0; hh < 2; hh++) {
					const int n = 2 * pr + hh, ci = n < 4 ? n : n + 1; // 3x3 raster without the centre
					const int vx = cx + (ci % 3 - 1) * step, vy = cy + (ci / 3 - 1) * step;
					packed |= __builtin_amdgcn_sad_u8(pred4(vx, vy), cur4, 0u) << (16 * hh);
				}
				const uint32_t tot = (uint32_t)wave_sum_i32((int)packed);
#pragma unroll
				for (int hh = 0; hh < 2; hh++) {
					const int n = 2 * pr + hh, ci = n < 4 ? n : n + 1;
					const int vx = cx + (ci % 3 - 1) * step, vy = cy + (ci / 3 - 1) * step;
					const uint32_t j = ((tot >> (16 * hh)) & 0xffffu) + (uint32_t)(qs.lambda * (mv_bits_q(vx) + mv_bits_q(vy)));
					const uint32_t k2 = (j << 4) | (uint32_t)(n + 1);
					key = (allowed(vx, vy) && k2 < key) ? k2 : key;
				}
			}
			const int win = (int)(key & 15u);
			if (win) {
				const int n = win - 1, ci = n < 4 ? n : n + 1;
				qx = cx + (ci % 3 - 1) * step;
				qy = cy + (ci / 3 - 1) * step;
				bestj = key >> 4;
			}
		}
		final_cost = bestj;
	}
	// scene cuts (Params::cut_votes): does the macroblock's own mean predict it better than the best vector? The
	// session counts the votes of the picture and codes it again as an IDR picture if they reach its threshold.
	if (p.cut_votes && !(REFRESH && refresh)) {
		const uint32_t mine = ((const uint32_t *)s_cur)[lane];
		const int mean = (wave_sum_i32((int)__builtin_amdgcn_sad_u8(mine, 0u, 0u)) + 128) >> 8;
		const int sadm = wave_sum_i32((int)__builtin_amdgcn_sad_u8(mine, (uint32_t)mean * 0x01010101u, 0u));
		if (lane == 0 && valid && (uint32_t)(sadm + 32 * qs.lambda) < final_cost)
			atomicAdd(p.cut_votes, 1u);
	}
	// MAD classes (Params::mad_on): the SAD of the chosen prediction -- its cost without the vector's
	// share -- picks one of five QP offsets for the residual; the decision above was taken without it
	if (MBQP) {
		int qp_mb = qp_base;
		if (p.mad_on) {
			int sad = (int)final_cost - qs.lambda * (mv_bits_q(qx) + mv_bits_q(qy));
			if (REFRESH && refresh) // against the constant prediction
				sad = wave_sum_i32((int)__builtin_amdgcn_sad_u8(((const uint32_t *)s_cur)[lane], 0x80808080u, 0u));
			qp_mb = mad_class_qp(p, qp_base, sad);
			qs = qp_set<true>(p, qp_mb);
		}
		if (lane == 0 && valid)
			p.qp_t[mbaddr] = (uint8_t)qp_mb;
	}

	// ---- residual, transform, quantisation, reconstruction: one lane per SAMPLE ----
	// 16-lane group = one 4x4 block, transforms across lanes with DPP (enc_device.h):
	// 4 passes cover the 16 luma blocks, 2 passes the 8 chroma blocks, all 64 lanes busy.
	for (int i = lane; i < COEF_PER_MB * 2 / 16; i += 64)
		((u32x4 *)s_coef)[i] = u32x4{ 0, 0, 0, 0 };
	WAVE_LDS_SYNC();

	const int f_inter = (1 << qs.qbits) / 6, f_inter_c = (1 << qs.qbits_c) / ((REFRESH && refresh) ? 3 : 6);
	const int gq = lane >> 4, l = lane & 15;
	const int pi = l >> 2, pj = l & 3;
	const LaneK ci = lane_consts(pi), cj = lane_consts(pj);
	const int kv = pi == 1 ? 2 : pi == 2 ? 1 : pi, kh = pj == 1 ? 2 : pj == 2 ? 1 : pj;
	const int rpos = 4 * kv + kh;             // coefficient (raster) this lane ends up holding
	const int pclass = (kv & 1) == (kh & 1) ? (kv & 1) : 2;
	const int zz = zigzag_of(rpos);
	const int mf_y = pclass == 0 ? qs.mf[0] : pclass == 1 ? qs.mf[1] : qs.mf[2];
	const int dv_y = pclass == 0 ? qs.dv[0] : pclass == 1 ? qs.dv[1] : qs.dv[2];
	const int mf_c = pclass == 0 ? qs.mfc[0] : pclass == 1 ? qs.mfc[1] : qs.mfc[2];
	const int dv_c = pclass == 0 ? qs.dvc[0] : pclass == 1 ? qs.dvc[1] : qs.dvc[2];

	// (counts go to LDS from every lane of a block -- same value, same address, no branch -- and
	// leave as three 8-byte stores at the end)
	uint32_t nz_luma = 0; // bit = coding-order block index
	int n_levels = 0;     // levels of the macroblock that are not zero (wave-uniform)
	uint64_t big_levels = 0; // lanes that have seen a level that does not fit a byte (wave-uniform)
	const uint8_t *predp = s_ref + (16 + bdy) * REF_STRIDE + 16 + bdx;
	// fractional vector: the prediction is the rounded mean of two plane samples (kSubTab)
	const bool frac = SUBPEL && ((qx | qy) & 3) != 0;
	const uint8_t *pa_base = predp, *pb_base = predp;
	int pa_stride = REF_STRIDE, pb_stride = REF_STRIDE;
	if (frac) {
		const SubPick k = kSubTab[4 * (qy & 3) + (qx & 3)];
		const int ox = (qx >> 2) - bdx, oy = (qy >> 2) - bdy;
		pa_base = k.pa ? s_sub + (k.pa - 1) * 18 * SP_STRIDE + (oy + k.va + 1) * SP_STRIDE + ox + k.ua + 1
			       : predp + (oy + k.va) * REF_STRIDE + ox + k.ua;
		pb_base = k.pb ? s_sub + (k.pb - 1) * 18 * SP_STRIDE + (oy + k.vb + 1) * SP_STRIDE + ox + k.ub + 1
			       : predp + (oy + k.vb) * REF_STRIDE + ox + k.ub;
		pa_stride = k.pa ? SP_STRIDE : REF_STRIDE;
		pb_stride = k.pb ? SP_STRIDE : REF_STRIDE;
	}
	if (REFRESH && refresh) {
		// Intra16x16, prediction 128 (8.3.3 with no neighbour): the 16 DC coefficients go through the
		// 4x4 Hadamard (8.5.10), the other 15 of a block are quantised with the intra dead zone
		const int f_intra = (1 << qs.qbits) / 3;
		int wl[4], dcm[16];
#pragma unroll
		for (int pass = 0; pass < 4; pass++) {
			const int idx = 4 * pass + gq, bx = blk_x(idx), by = blk_y(idx);
			wl[pass] = fdct_lanes24((int)s_cur[(4 * by + pi) * 16 + 4 * bx + pj] - 128, ci, cj);
#pragma unroll
			for (int k = 0; k < 4; k++) // lane (0,0) of a group holds its DC: raster position of block 4*pass+k
				dcm[blk_x(4 * pass + k) + 4 * blk_y(4 * pass + k)] = __builtin_amdgcn_readlane(wl[pass], 16 * k);
		}
		// Hadamard rows (1,1,1,1) (1,1,-1,-1) (1,-1,-1,1) (1,-1,1,-1): sign bit of row k at element r is
		// bit 4k+r of 0xA6C0. Lane l < 16 makes output (k2 = l / 4, k1 = l % 4).
		const uint32_t SB = 0xA6C0u;
		const int k2 = (lane >> 2) & 3, k1 = lane & 3;
		int hsum = 0;
#pragma unroll
		for (int r = 0; r < 4; r++)
#pragma unroll
			for (int c = 0; c < 4; c++)
				hsum += (((SB >> (4 * k2 + r)) ^ (SB >> (4 * k1 + c))) & 1u) ? -dcm[4 * r + c] : dcm[4 * r + c];
		const int dlev = quant1((hsum + 1) >> 1, qs.mf[0], 2 * f_intra, qs.qbits + 1);
		if (lane < 16)
			s_coef[COEF_DC16 + zigzag_of(lane)] = (int16_t)dlev;
		int cm[16];
#pragma unroll
		for (int r = 0; r < 16; r++)
			cm[r] = __builtin_amdgcn_readlane(dlev, r);
		int isum = 0; // lane l < 16: the scaled DC of the block at (bx = l % 4, by = l / 4)
#pragma unroll
		for (int r = 0; r < 4; r++)
#pragma unroll
			for (int c = 0; c < 4; c++)
				isum += (((SB >> (4 * k2 + r)) ^ (SB >> (4 * k1 + c))) & 1u) ? -cm[4 * r + c] : cm[4 * r + c];
		const int ls = 16 * qs.dv[0];
		const int dcr = qs.qshift >= 6 ? (isum * ls) << (qs.qshift - 6) : (isum * ls + (1 << (5 - qs.qshift))) >> (6 - qs.qshift);
		bool any_ac = false;
#pragma unroll
		for (int pass = 0; pass < 4; pass++) {
			const int idx = 4 * pass + gq, bx = blk_x(idx), by = blk_y(idx);
			const int x = 4 * bx + pj, y = 4 * by + pi;
			const int level = l == 0 ? 0 : quant24(wl[pass], mf_y, f_intra, qs.qbits);
			s_coef[COEF_LUMA + idx * 16 + zz] = (int16_t)level;
			const int mydc = __shfl(dcr, bx + 4 * by);
			const int rr = idct_lanes24(l == 0 ? mydc : __mul24(level, dv_y) << qs.qshift, ci, cj);
			s_recy[y * 16 + x] = (uint8_t)clip255(128 + rr);
			const uint64_t bal = __ballot(level != 0);
			s_nnz[bx + 4 * by] = (uint8_t)__popc((uint32_t)(bal >> (16 * gq)) & 0xffffu);
			any_ac = any_ac || bal != 0;
		}
		nz_luma = any_ac ? 0xffffu : 0u; // Intra16x16: the luma part of the coded block pattern is 0 or 15
	} else {
#pragma unroll
	for (int pass = 0; pass < 4; pass++) {
		const int idx = 4 * pass + gq, bx = blk_x(idx), by = blk_y(idx);
		const int x = 4 * bx + pj, y = 4 * by + pi;
		const int pv = frac ? ((int)pa_base[y * pa_stride + x] + (int)pb_base[y * pb_stride + x] + 1) >> 1
				    : (int)predp[y * REF_STRIDE + x];
		const int res = (int)s_cur[y * 16 + x] - pv;
		const int w = fdct_lanes24(res, ci, cj);
		const int level = quant24(w, mf_y, f_inter, qs.qbits);
		s_coef[COEF_LUMA + idx * 16 + zz] = (int16_t)level;
		const int rr = idct_lanes24(__mul24(level, dv_y) << qs.qshift, ci, cj);
		s_recy[y * 16 + x] = (uint8_t)clip255(pv + rr);
		const uint64_t bal = __ballot(level != 0);
		big_levels |= __ballot(level > 127 || level < -127);
		n_levels += __popcll(bal); // (scalar: a count for the I_PCM gate below)
		s_nnz[bx + 4 * by] = (uint8_t)__popc((uint32_t)(bal >> (16 * gq)) & 0xffffu);
#pragma unroll
		for (int k = 0; k < 4; k++)
			nz_luma |= ((uint32_t)(bal >> (16 * k)) & 0xffffu) ? (1u << (4 * pass + k)) : 0u;
	}
	}
	// chroma: pass = plane, group = block (raster 2x2)
	// vector in 1/8 chroma samples = the luma vector in quarter samples (8.4.1.4)
	const int ix = qx >> 3, iy = qy >> 3, fx = qx & 7, fy = qy & 7;
	int cpv[2], cdq[2], dcw[2][4];
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
		if (REFRESH && refresh)
			cpv[c] = 128;
		const int res = (int)s_curc[y * 16 + 2 * x + c] - cpv[c];
		const int w = fdct_lanes24(res, ci, cj);
		// lane (0,0) of a group holds its DC coefficient: the four of a plane become wave-uniform
#pragma unroll
		for (int k = 0; k < 4; k++)
			dcw[c][k] = __builtin_amdgcn_readlane(w, 16 * k);
		const int level = l == 0 ? 0 : quant24(w, mf_c, f_inter_c, qs.qbits_c);
		s_coef[COEF_CAC + (4 * c + gq) * 16 + zz] = (int16_t)level; // (entry 0 of a block is unused: lane 0 writes its 0 there)
		cdq[c] = __mul24(level, dv_c) << qs.qshift_c;
		const uint64_t bal = __ballot(level != 0);
		big_levels |= __ballot(level > 127 || level < -127);
		n_levels += __popcll(bal);
		s_nnz[16 + 4 * c + gq] = (uint8_t)__popc((uint32_t)(bal >> (16 * gq)) & 0xffffu);
		nz_cac |= bal ? 1u : 0u;
	}
	// 2x2 Hadamard of the chroma DCs on lanes 0..7 = (plane, output index); the quantised four of a
	// plane become wave-uniform again for the inverse
	int lqv;
	{
		const int c = (lane >> 2) & 1, blk = lane & 3;
		const int d0c = c ? dcw[1][0] : dcw[0][0], d1c = c ? dcw[1][1] : dcw[0][1];
		const int d2c = c ? dcw[1][2] : dcw[0][2], d3c = c ? dcw[1][3] : dcw[0][3];
		const int s1 = (blk & 1) ? -1 : 1, s2 = (blk & 2) ? -1 : 1;
		const int hd = d0c + s1 * d1c + s2 * d2c + s1 * s2 * d3c;
		lqv = quant1(hd, qs.mfc[0], 2 * f_inter_c, qs.qbits_c + 1);
	}
	if (lane < 8)
		s_coef[COEF_CDC + lane] = (int16_t)lqv;
	const uint64_t dcmask = __ballot(lqv != 0) & 0xffull;
	big_levels |= __ballot(lqv > 127 || lqv < -127) & 0xffull;
	n_levels += __popcll(dcmask);
	int lqs[8];
#pragma unroll
	for (int k = 0; k < 8; k++)
		lqs[k] = __builtin_amdgcn_readlane(lqv, k);
#pragma unroll
	for (int c = 0; c < 2; c++) {
		const int bx = gq & 1, by = gq >> 1;
		const int x = 4 * bx + pj, y = 4 * by + pi;
		const int s1 = (gq & 1) ? -1 : 1, s2 = (gq & 2) ? -1 : 1;
		const int fd = lqs[4 * c] + s1 * lqs[4 * c + 1] + s2 * lqs[4 * c + 2] + s1 * s2 * lqs[4 * c + 3];
		const int dcr = ((fd * qs.dvc[0] * 16) << qs.qshift_c) >> 5;
		const int rr = idct_lanes24(l == 0 ? dcr : cdq[c], ci, cj);
		s_recc[y * 16 + 2 * x + c] = (uint8_t)clip255(cpv[c] + rr);
	}
	// coded block pattern
	int cbp = 0;
#pragma unroll
	for (int qd = 0; qd < 4; qd++)
		cbp |= ((nz_luma >> (4 * qd)) & 0xfu) ? (1 << qd) : 0;
	cbp |= nz_cac ? 0x20 : (dcmask ? 0x10 : 0);
	// macroblocks differ in QP: the first one of a row always carries an mb_qp_delta ("chroma DC" with two
	// empty blocks when it has no coded block; it is then no P_Skip either), so that QP_Y is never
	// inherited across a row boundary and deblocking can follow the chain left to right
	if (MBQP && p.row_rule && mx == 0 && !(REFRESH && refresh) && cbp == 0)
		cbp = 0x10;
	WAVE_LDS_SYNC();
	// I_PCM (Params::pcm; H.264 A.3.1 allows macroblock_layer() 3200 bits): the levels are final here, so the
	// bits CAVLC will spend on them are known but for coeff_token's table (cavlc_block_len takes the longest)
	// and the header (PCM_HEADER_BITS covers it). Fewer than 64 levels cannot get there -- 26 blocks x 16 bits of
	// coeff_token + 9 of total_zeros, 28 + 3 at most per level: the rare path of a rare picture. If the count
	// passes the limit the macroblock is its source samples: reconstruction, 16 coefficients counted in every
	// block for the neighbours' nC (9.2.1), no level, no vector (intra).
	bool is_pcm = false;
	if (!REFRESH && !MBQP && p.pcm && n_levels >= 64) {
		int bits = 0;
		if (lane < 16) {
			if (cbp & (1 << (lane >> 2)))
				bits = cavlc_block_len(s_coef + COEF_LUMA + 16 * lane, 16, false);
		} else if (lane < 18) {
			if (cbp & 0x30)
				bits = cavlc_block_len(s_coef + COEF_CDC + 4 * (lane - 16), 4, true);
		} else if (lane < 26) {
			if (cbp & 0x20)
				bits = cavlc_block_len(s_coef + COEF_CAC + 16 * (lane - 18) + 1, 15, false);
		}
		is_pcm = wave_sum_i32(bits) + PCM_HEADER_BITS > MB_BITS_LIMIT;
		if (is_pcm) {
			if (lane < 16)
				*(u32x4 *)(s_recy + lane * 16) = *(const u32x4 *)(s_cur + lane * 16);
			else if (lane < 24)
				*(u32x4 *)(s_recc + (lane - 16) * 16) = *(const u32x4 *)(s_curc + (lane - 16) * 16);
			else if (lane < 48)
				s_nnz[lane - 24] = 16;
			if (lane < COEF_PER_MB * 2 / 16)
				((u32x4 *)s_coef)[lane] = u32x4{ 0, 0, 0, 0 };
			cbp = 0;
			qx = qy = 0;
			WAVE_LDS_SYNC();
		}
	}

	// ---- write back: levels (51 x 16 B, or 26 x 16 B as bytes), reconstruction rows, records ----
	// levels: only the block groups the coded block pattern names (nobody reads the others, and at the rates of real
	// pictures most are empty: 6.7 MB per 1080p picture were written, then read) -- and as BYTES when every level of the
	// macroblock fits one (MB_LEVELS_INT8: the same indices, half the bytes: -1.6 MB written, -3.4 MB read per 1080p P picture
	// at QP 24). A refresh macroblock keeps int16 (its Intra16x16 DC levels are the large ones).
	const bool levels8 = !(REFRESH && refresh) && !is_pcm && cbp != 0 && big_levels == 0;
	if (levels8) {
		u32x4 a = u32x4{ 0, 0, 0, 0 }, b = a;
		if (lane < 26) { // level indices 16 lane .. 16 lane + 15 (the last chunk reads 8 int16 of slack behind the 408: the arena is larger)
			a = *(const u32x4 *)((const uint8_t *)s_coef + lane * 32);
			b = *(const u32x4 *)((const uint8_t *)s_coef + lane * 32 + 16);
		}
		WAVE_LDS_SYNC();
		if (lane < 26) {
			auto pk = [](uint32_t w0, uint32_t w1) { // four int16 -> four int8
				return (w0 & 0xffu) | ((w0 >> 8) & 0xff00u) | ((w1 & 0xffu) << 16) | ((w1 << 8) & 0xff000000u);
			};
			*(u32x4 *)((uint8_t *)s_coef + lane * 16) = u32x4{ pk(a.x, a.y), pk(a.z, a.w), pk(b.x, b.y), pk(b.z, b.w) };
		}
		WAVE_LDS_SYNC();
		bool need;
		if (lane < 16)
			need = (cbp >> (lane >> 2)) & 1;   // luma block `lane`
		else if (lane == 17)
			need = (cbp & 0x30) != 0;           // chroma DC (and the first 8 levels of the first chroma AC block)
		else
			need = lane >= 18 && lane < 26 && (cbp & 0x20) != 0;
		if (need)
			*(u32x4 *)((uint8_t *)(p.coef + mbaddr * COEF_PER_MB) + lane * 16) = *(const u32x4 *)((const uint8_t *)s_coef + lane * 16);
	} else {
		bool need;
		if (lane < 32)
			need = (cbp >> (lane >> 3)) & 1;
		else if (lane < 34)
			need = REFRESH && refresh;
		else if (lane == 34)
			need = (cbp & 0x30) != 0;
		else
			need = lane < COEF_PER_MB * 2 / 16 && (cbp & 0x20) != 0;
		if (need)
			*(u32x4 *)((uint8_t *)(p.coef + mbaddr * COEF_PER_MB) + lane * 16) = *(const u32x4 *)((const uint8_t *)s_coef + lane * 16);
	}
	if (lane < 16)
		*(u32x4 *)(rec_y + (uint32_t)lane * cw) = *(const u32x4 *)(s_recy + lane * 16);
	else if (lane < 24)
		*(u32x4 *)(rec_uv + (uint32_t)(lane - 16) * cw) = *(const u32x4 *)(s_recc + (lane - 16) * 16);
	if (lane < 3)
		*(uint64_t *)(p.nnz + mbaddr * NNZ_PER_MB + 8 * lane) = *(const uint64_t *)(s_nnz + 8 * lane);
	if (lane == 4)
		*(u32x4 *)(p.i4modes + mbaddr * 16) = u32x4{ 0x02020202u, 0x02020202u, 0x02020202u, 0x02020202u };
	if (lane == 0) {
		MbInfo m;
		m.type = (REFRESH && refresh) ? MB_I16x16 : is_pcm ? MB_IPCM : MB_P16x16;
		m.i16_mode = (REFRESH && refresh) ? 2 : 0; // DC
		m.chroma_mode = levels8 ? (int8_t)MB_LEVELS_INT8 : 0; // DC; an inter macroblock: how its levels are stored
		m.cbp = (int8_t)cbp;
		m.mvx = (int16_t)qx; // quarter samples
		m.mvy = (int16_t)qy;
		p.mbi[mbaddr] = m;
	}
	// tail_gate: one wave holds the launch open until the NEXT stripe may start (what k_gate in front of that stripe would
	// have waited for, without its launch): a launch ends when its last wave does
	if (p.gate_next_need > 0 && blockIdx.x == 0 && wave == 0) {
		if (lane == 0)
			dbg_event(p, 10, p.db_tag, (uint32_t)p.gate_next_need);
		wait_prev_deblock(p, lane, p.gate_next_need);
		if (lane == 0)
			dbg_event(p, 11, p.db_tag, (uint32_t)p.gate_next_need);
	}
}

// ------------------------------------------------------------------------
// k_qp_eff (sessions whose macroblocks differ in QP): QP_Y of every macroblock as the decoder will
// hold it, for the mb_qp_delta of the one after it (the deblock kernel follows the same chain itself,
// row by row). One workgroup per unit (stream x slice), after the unit's analysis, on the entropy
// stream. A macroblock changes QP_Y only if it carries mb_qp_delta --
// Intra16x16, or any coded block (7.3.5, 7.4.5); the others, P_Skip included, keep that of the
// macroblock before them in decoding order (the slice's at its start): a max-scan over "index of
// the last macroblock with an mb_qp_delta".
// ------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_qp_eff(Params p)
{
	__shared__ int s_wave[16];
	__shared__ int s_carry;
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int unit = blockIdx.x, st = unit / p.slice_num, sl = p.slice_first + unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, sl, &r0, &r1);
	const int n = (r1 - r0) * p.mb_w;
	const size_t base = (size_t)st * p.n_mb + (size_t)r0 * p.mb_w;
	if (tid == 0)
		s_carry = -1;
	__syncthreads();
	for (int i0 = 0; i0 < n; i0 += 1024) {
		const int i = i0 + tid;
		int v = -1;
		if (i < n) {
			const MbInfo m = p.mbi[base + i];
			if (m.type == MB_I16x16 || (m.cbp & 0x3f))
				v = i;
		}
#pragma unroll
		for (int off = 1; off < 64; off <<= 1) {
			const int o = __shfl_up(v, off);
			if (lane >= off)
				v = max(v, o);
		}
		if (lane == 63)
			s_wave[wave] = v;
		__syncthreads();
		int before = s_carry;
		for (int w = 0; w < wave; w++)
			before = max(before, s_wave[w]);
		v = max(v, before);
		if (i < n) {
			p.qp_eff[base + i] = (uint8_t)(v >= 0 ? (int)p.qp_t[base + v] : pic_qp(p, st));
		}
		__syncthreads(); // (everybody has read s_carry and s_wave)
		if (tid == 1023)
			s_carry = v;
		__syncthreads();
	}
}

int Launch::qp_eff(const Params &p, void *stream)
{
	hipLaunchKernelGGL(k_qp_eff, dim3(p.n_streams * p.slice_num), dim3(1024), 0, (hipStream_t)stream, p);
	return (int)hipGetLastError();
}

int Launch::inter(const Params &p, void *stream, int n_mbs)
{
	const dim3 grid((n_mbs + INTER_WAVES - 1) / INTER_WAVES), block(64 * INTER_WAVES);
	// the refinement's planes take LDS, the refresh branch and the per-macroblock QP registers: kernels
	// of their own, so the whole-sample, all-inter, one-QP one keeps its occupancy
	const int form = (p.subpel > 0 ? 1 : 0) | (p.intra_refresh > 0 ? 2 : 0) | (p.roi_off ? 4 : 0);
	switch (form) {
	case 0: hipLaunchKernelGGL((k_inter<0, false, false>), grid, block, 0, (hipStream_t)stream, p); break;
	case 1: hipLaunchKernelGGL((k_inter<1, false, false>), grid, block, 0, (hipStream_t)stream, p); break;
	case 2: hipLaunchKernelGGL((k_inter<0, true, false>), grid, block, 0, (hipStream_t)stream, p); break;
	case 3: hipLaunchKernelGGL((k_inter<1, true, false>), grid, block, 0, (hipStream_t)stream, p); break;
	case 4: hipLaunchKernelGGL((k_inter<0, false, true>), grid, block, 0, (hipStream_t)stream, p); break;
	case 5: hipLaunchKernelGGL((k_inter<1, false, true>), grid, block, 0, (hipStream_t)stream, p); break;
	case 6: hipLaunchKernelGGL((k_inter<0, true, true>), grid, block, 0, (hipStream_t)stream, p); break;
	default: hipLaunchKernelGGL((k_inter<1, true, true>), grid, block, 0, (hipStream_t)stream, p); break;
	}
	return (int)hipGetLastError();
}

// ------------------------------------------------------------------------
// k_gate: the only place a frame WAITS for the frame before it. One wave. A P picture is
// analysed in a few diagonal stripes; in front of stripe [s0, s1) this kernel (1) publishes
// how far this frame's analysis has come (the deblock kernel of this frame, already running
// on another stream, polls that word) and (2) spins until every deblock band of the previous
// frame has written back the macroblocks with x + y < need_d (the deblock kernel's own measure of
// progress: one macroblock of lag per row): the motion search of (x, y) reads the final samples of
// (x-1..x+1, y-1..y+1); a macroblock is final once its right and its lower neighbour are filtered and
// written back, for (x+1, y+1) those are on x + y + 3, i.e. need_d = s1 + 3 for the stripe's last
// diagonal s1 - 1 (one more is asked for).
// Spins are bounded: a time-out leaves an error code and lets the stream drain.
// ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_gate(Params p, int publish_d, int need_d)
{
	const int lane = threadIdx.x;
	if (lane == 0)
		dbg_event(p, 7, p.db_tag, (uint32_t)(publish_d & 0xffff) | (uint32_t)need_d << 16);
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
			// a band owns the diagonals x + y from r0 + 4*band on: nothing of it is below that
			pending = band < ((r1 - r0 + 3) >> 2) && r0 + 4 * band < need_d;
		}
		uint32_t spins = 0;
		uint64_t t0 = 0;
		while (true) {
			if (pending) {
				const uint32_t v = __hip_atomic_load(&p.prev_db[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				pending = (int32_t)(v - (p.prev_tag | (uint32_t)need_d)) < 0;
			}
			if (!__any(pending))
				break;
			// (the session's abort word is looked at every 64th spin: a second load per look would stretch the hand-off)
			if (wait_expired(++spins, t0) || ((spins & 63u) == 0 && __hip_atomic_load(p.abort_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { // wall time, not iterations
				if (lane == 0) {
					__hip_atomic_store(p.wf_err, 8u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					__hip_atomic_store(p.abort_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
				}
				return;
			}
			__builtin_amdgcn_s_sleep(32);
		}
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
	if (lane == 0)
		dbg_event(p, 8, p.db_tag, (uint32_t)need_d);
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
	if (p.cut_votes && blockIdx.x == 0 && (int)threadIdx.x == p.n_streams + 2) { // scene cuts (one stream, one cohort): the picture's votes
		host_sizes[threadIdx.x] = p.cut_votes[0];                              // (the analysis is over: the entropy stream waited for it)
		p.cut_votes[0] = 0;                                                    // ... and the slot's next P picture starts from zero
	}
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
	__shared__ uint8_t s_best4[16];        // I_PCM estimate: the chosen Intra4x4 mode of every block
	__shared__ __attribute__((aligned(16))) int16_t s_lv[24][16]; // ... and the open loop's levels, zig-zag

	const int lane = threadIdx.x;
	if (blockIdx.x == 0 && lane == 0)
		dbg_event(p, 2, p.db_tag, 0);
	if (p.src_direct && blockIdx.x == 0 && (p.wf_split_intra > 1 || p.ir_rows)) {
		// (what k_input does where it runs: the ticket / abort / progress words of the intra wavefront start at zero)
		const int used = 16 + p.n_streams * p.slice_num * WF_MAX_SPLIT;
		for (int i = lane; i < used; i += 64)
			p.wf_sync[i] = 0;
	}
	int mx, my, st;
	xcd_mb(p, mx, my, st);
	const int x0 = mx * 16, y0 = my * 16;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, slice_of_row(p.mb_h, p.slice_count, my), &r0, &r1);
	const bool top_mb = my > r0, left_mb = mx > 0, tr_mb = top_mb && mx < p.mb_w - 1;
	const uint8_t *src_y = src_luma(p, st);
	const uint8_t *src_uv = src_chroma(p, st);
	const size_t mbaddr = (size_t)st * p.n_mb + (size_t)my * p.mb_w + mx;

	// load tiles (byte loads; out-of-picture entries are never used)
	for (int i = lane; i < 17 * 21; i += 64) {
		int ty = i / 21, tx = i - ty * 21;
		int gx = x0 - 1 + tx, gy = y0 - 1 + ty;
		uint8_t v = 128;
		if (gx >= 0 && gy >= 0 && gx < p.cw && gy < p.ch)
			v = src_y[(size_t)min(gy, p.src_last_y) * p.cw + gx];
		s_t[ty * TS + tx] = v;
	}
	for (int i = lane; i < 2 * 81; i += 64) {
		int c = i / 81, j = i - c * 81, ty = j / 9, tx = j - ty * 9;
		int gx = mx * 8 - 1 + tx, gy = my * 8 - 1 + ty;
		uint8_t v = 128;
		if (gx >= 0 && gy >= 0 && gx < (p.cw >> 1) && gy < (p.ch >> 1))
			v = src_uv[(size_t)min(gy, p.src_last_c) * p.cw + 2 * gx + c];
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
	int dc16 = 128, pl16_a = 0, pl16_b = 0, pl16_c = 0; // (kept for the I_PCM estimate below)
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
		dc16 = dcv; pl16_a = a; pl16_b = b; pl16_c = cc;
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

	// I_PCM (Params::pcm), decided open loop like everything here: the residual of the CHOSEN predictions from
	// source samples, every 4x4 block -- 16 luma, 4 + 4 chroma -- through the plain transform and the intra
	// quantiser (no DC Hadamards), the zig-zag levels through CAVLC's length (cavlc_block_len). The closed loop's
	// levels differ a little: PCM_LIMIT_OPEN leaves a tenth of room below the 3200 bits of H.264 A.3.1.
	bool is_pcm = false;
	if (p.pcm) {
		const bool is4 = sum4 + 32 * p.lambda < (int)(best16 >> 2);
		const int m16 = (int)(best16 & 3u), mc = (int)(bestc & 3u);
		if (lane < 16)
			s_best4[lane] = (uint8_t)best4;
		__syncthreads();
		const int gq = lane >> 4, l = lane & 15, pi = l >> 2, pj = l & 3;
		const LaneK ci = lane_consts(pi), cj = lane_consts(pj);
		const int kv = pi == 1 ? 2 : pi == 2 ? 1 : pi, kh = pj == 1 ? 2 : pj == 2 ? 1 : pj;
		const int rpos = 4 * kv + kh, pclass = (kv & 1) == (kh & 1) ? (kv & 1) : 2, zz = zigzag_of(rpos);
		const int mf_y = pclass == 0 ? p.mf[0] : pclass == 1 ? p.mf[1] : p.mf[2];
		const int mf_c = pclass == 0 ? p.mfc[0] : pclass == 1 ? p.mfc[1] : p.mfc[2];
		const int f_y = (1 << p.qbits) / 3, f_c = (1 << p.qbits_c) / 3;
#pragma unroll
		for (int pass = 0; pass < 4; pass++) {
			const int idx = 4 * pass + gq, bx = blk_x(idx), by = blk_y(idx);
			const int x = 4 * bx + pj, y = 4 * by + pi;
			int pv;
			if (is4)
				pv = pred4_px(p.pred4_tab[(int)s_best4[idx] * 16 + l], s_E[idx], s_dcv[idx]);
			else
				pv = m16 == 0 ? (int)s_t[1 + x] : m16 == 1 ? (int)s_t[(1 + y) * TS] : m16 == 2 ? dc16
				   : clip255((pl16_a + pl16_b * (x - 7) + pl16_c * (y - 7) + 16) >> 5);
			const int w = fdct_lanes24((int)s_t[(1 + y) * TS + 1 + x] - pv, ci, cj);
			s_lv[idx][zz] = (int16_t)quant24(w, mf_y, f_y, p.qbits);
		}
#pragma unroll
		for (int c = 0; c < 2; c++) {
			const uint8_t *T = s_c[c];
			const int bx = gq & 1, by = gq >> 1, x = 4 * bx + pj, y = 4 * by + pi;
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
			const int pv = mc == 0 ? dcv : mc == 1 ? (int)T[(1 + y) * 12] : mc == 2 ? (int)T[1 + x]
				     : clip255((a + b * (x - 3) + cc * (y - 3) + 16) >> 5);
			const int w = fdct_lanes24((int)T[(1 + y) * 12 + 1 + x] - pv, ci, cj);
			s_lv[16 + 4 * c + gq][zz] = (int16_t)quant24(w, mf_c, f_c, p.qbits_c);
		}
		__syncthreads();
		const int bits = lane < 24 ? cavlc_block_len(s_lv[lane], 16, false) : 0;
		is_pcm = wave_sum_i32(bits) + PCM_HEADER_BITS > PCM_LIMIT_OPEN;
	}
	if (lane < 16)
		p.i4modes[mbaddr * 16 + lane] = (int8_t)(is_pcm ? 2 : best4);
	if (lane == 0) {
		MbInfo m;
		const int qp_base = p.roi_off ? roi_qp(p, st, mx, my) : p.qp;
		m.type = (sum4 + 32 * (p.roi_off ? p.qtab[qp_base].lambda : p.lambda) < (int)(best16 >> 2)) ? MB_I4x4 : MB_I16x16;
		if (p.roi_off) // the QP the macroblock is quantised with: its region's, moved by its MAD class (SAD of the chosen kind of prediction)
			p.qp_t[mbaddr] = (uint8_t)(p.mad_on ? mad_class_qp(p, qp_base, m.type == MB_I4x4 ? sum4 : (int)(best16 >> 2)) : qp_base);
		m.i16_mode = (int8_t)(best16 & 3u);
		m.chroma_mode = (int8_t)(bestc & 3u);
		if (is_pcm) {
			m.type = MB_IPCM;
			m.i16_mode = 0;
			m.chroma_mode = 0;
		}
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
