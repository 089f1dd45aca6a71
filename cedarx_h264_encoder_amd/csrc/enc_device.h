// cedarx_h264_encoder_amd/csrc/enc_device.h
// Device-side building blocks shared by the gfx950 kernels: 4x4 integer
// transform pair, quantiser, wave reductions, intra predictors.
// Everything is integer arithmetic (H.264 clause 8.5); results must equal the
// scalar oracle bit for bit, which tests/ check through the C-ABI.
#pragma once

#include <hip/hip_runtime.h>

#include "enc_internal.h"

namespace cedar {

#define DEVI __device__ __forceinline__

constexpr int LEVEL_MAX = 2047; // keeps every level inside the baseline CAVLC escape range

// zig-zag scan index -> raster index, and position class of a raster index
__device__ constexpr uint8_t kZigzag[16] = { 0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15 };
__device__ constexpr uint8_t kPosClass[16] = { 0, 2, 0, 2, 2, 1, 2, 1, 0, 2, 0, 2, 2, 1, 2, 1 };
// coding-order 4x4 block index -> position in 4x4-block units
__device__ constexpr uint8_t kBlkX[16] = { 0, 1, 0, 1, 2, 3, 2, 3, 0, 1, 0, 1, 2, 3, 2, 3 };
__device__ constexpr uint8_t kBlkY[16] = { 0, 0, 1, 1, 0, 0, 1, 1, 2, 2, 3, 3, 2, 2, 3, 3 };

// arithmetic forms of kBlkX / kBlkY (no table load on a dependent chain)
DEVI int blk_x(int idx) { return ((idx >> 1) & 2) | (idx & 1); }
DEVI int blk_y(int idx) { return ((idx >> 2) & 2) | ((idx >> 1) & 1); }

DEVI int clip255(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }
DEVI int iabs(int v) { return v < 0 ? -v : v; }
DEVI int clip3(int lo, int hi, int v) { return v < lo ? lo : (v > hi ? hi : v); }

DEVI uint32_t wave_min_u32(uint32_t v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		uint32_t o = (uint32_t)__shfl_xor((int)v, off, 64);
		v = o < v ? o : v;
	}
	return v;
}

DEVI int wave_sum_i32(int v)
{
#pragma unroll
	for (int off = 32; off > 0; off >>= 1)
		v += __shfl_xor(v, off, 64);
	return v;
}

// forward core transform, raster in / raster out (clause 8.5 encoder twin)
DEVI void fdct4x4(const int r[16], int w[16])
{
	int t[16];
#pragma unroll
	for (int i = 0; i < 4; i++) {
		int a = r[4 * i] + r[4 * i + 3], b = r[4 * i + 1] + r[4 * i + 2];
		int c = r[4 * i + 1] - r[4 * i + 2], d = r[4 * i] - r[4 * i + 3];
		t[4 * i] = a + b;
		t[4 * i + 1] = 2 * d + c;
		t[4 * i + 2] = a - b;
		t[4 * i + 3] = d - 2 * c;
	}
#pragma unroll
	for (int i = 0; i < 4; i++) {
		int a = t[i] + t[12 + i], b = t[4 + i] + t[8 + i];
		int c = t[4 + i] - t[8 + i], d = t[i] - t[12 + i];
		w[i] = a + b;
		w[4 + i] = 2 * d + c;
		w[8 + i] = a - b;
		w[12 + i] = d - 2 * c;
	}
}

// inverse core transform of scaled coefficients (rows first, then columns:
// clause 8.5.12.2), result (x+32)>>6 added to pred with clipping
DEVI void idct4x4_add(const int d[16], const int pred[16], int out[16])
{
	int t[16];
#pragma unroll
	for (int i = 0; i < 4; i++) {
		int e0 = d[4 * i] + d[4 * i + 2], e1 = d[4 * i] - d[4 * i + 2];
		int e2 = (d[4 * i + 1] >> 1) - d[4 * i + 3], e3 = d[4 * i + 1] + (d[4 * i + 3] >> 1);
		t[4 * i] = e0 + e3;
		t[4 * i + 1] = e1 + e2;
		t[4 * i + 2] = e1 - e2;
		t[4 * i + 3] = e0 - e3;
	}
#pragma unroll
	for (int i = 0; i < 4; i++) {
		int e0 = t[i] + t[8 + i], e1 = t[i] - t[8 + i];
		int e2 = (t[4 + i] >> 1) - t[12 + i], e3 = t[4 + i] + (t[12 + i] >> 1);
		out[i] = clip255(pred[i] + ((e0 + e3 + 32) >> 6));
		out[4 + i] = clip255(pred[4 + i] + ((e1 + e2 + 32) >> 6));
		out[8 + i] = clip255(pred[8 + i] + ((e1 - e2 + 32) >> 6));
		out[12 + i] = clip255(pred[12 + i] + ((e0 - e3 + 32) >> 6));
	}
}

DEVI int quant1(int w, int mf, int f, int qbits)
{
	uint32_t a = (uint32_t)iabs(w);
	int l = (int)((a * (uint32_t)mf + (uint32_t)f) >> qbits);
	l = l > LEVEL_MAX ? LEVEL_MAX : l;
	return w < 0 ? -l : l;
}

// Quantise w[] (raster) into lv[] (zig-zag) from scan position `first`, build the
// scaled coefficients d[] (raster; d[0] left alone when first == 1), count levels.
DEVI int quant_dequant(const int w[16], int first, const int32_t mf[3], const int32_t dv[3],
		       int f, int qbits, int qshift, int lv[16], int d[16])
{
	int n = 0;
#pragma unroll
	for (int i = 0; i < 16; i++) {
		if (i < first) {
			lv[i] = 0;
			continue;
		}
		const int r = kZigzag[i], c = kPosClass[r];
		int l = quant1(w[r], mf[c], f, qbits);
		lv[i] = l;
		n += l != 0;
		d[r] = (l * dv[c]) << qshift;
	}
	return n;
}

// ---- intra 4x4 prediction from the 15-entry edge array ----
// E[0]=E[1]=l3, E[2]=l2, E[3]=l1, E[4]=l0, E[5]=tl, E[6..13]=t0..t7, E[14]=t7
// tab entry = kind<<4 | k (built on the host by build_pred4_table)
DEVI int pred4_px(int ent, const uint8_t *E, int dc)
{
	int kind = ent >> 4, k = ent & 15;
	int a = E[k > 0 ? k - 1 : 0], b = E[k], c = E[k + 1 < 15 ? k + 1 : 14];
	int v = b;
	v = kind == 1 ? (b + c + 1) >> 1 : v;
	v = kind == 2 ? (a + 2 * b + c + 2) >> 2 : v;
	v = kind == 3 ? dc : v;
	return v;
}

DEVI bool mode4_allowed(int mode, bool has_top, bool has_left)
{
	switch (mode) {
	case 0: case 3: case 7: return has_top;
	case 1: case 8: return has_left;
	case 2: return true;
	default: return has_top && has_left;
	}
}

// availability of the neighbours of the 4x4 block with coding index idx
// (top-right by the decoding-order rule, clause 6.4.11.4)
DEVI void blk4_avail(int idx, bool top_mb, bool left_mb, bool tr_mb, bool &ht, bool &hl, bool &htr)
{
	const int bx = blk_x(idx), by = blk_y(idx);
	// internal top-right availability bitmask over coding index: 2,6,8,9,10,12,14
	const uint32_t tri = (1u << 2) | (1u << 6) | (1u << 8) | (1u << 9) | (1u << 10) | (1u << 12) | (1u << 14);
	ht = by > 0 || top_mb;
	hl = bx > 0 || left_mb;
	htr = by == 0 ? (bx < 3 ? top_mb : tr_mb) : ((tri >> idx) & 1);
}

// comparisons and selects as integer masks: `c ? a : b` chains make hipcc emit
// exec-mask branches, which the single scalar pipe of a CU pays for
DEVI int lt_mask(int a, int b) { return (a - b) >> 31; }          // a < b ? -1 : 0  (|a|,|b| < 2^30)
DEVI int nz_mask(int v) { return (v | -v) >> 31; }               // v != 0 ? -1 : 0
DEVI int sel(int m, int a, int b) { return b ^ ((a ^ b) & m); }  // m ? a : b for m in {0,-1}

// cross-lane exchanges inside a 16-lane DPP row; lane = 4*i + j
// (bound_ctrl: none of these patterns reads an invalid lane, and with it set the compiler need not
// materialise the "old" value -- a v_mov 0 in front of every exchange otherwise)
DEVI int dpp_xor1(int v) { return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true); } // quad_perm [1,0,3,2]
DEVI int dpp_xor3(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x1B, 0xf, 0xf, true); } // quad_perm [3,2,1,0]
DEVI int dpp_xor4(int v) { return dpp_xor3(__builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true)); } // row_half_mirror, j restored
DEVI int dpp_xor12(int v) { return dpp_xor3(__builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true)); } // row_mirror, j restored

// Per-lane constants of the butterflies below (k = position 0..3 along the axis).
// Selecting with `k == 0 ? .. : k == 1 ? ..` makes hipcc emit exec-mask if/else
// chains (half of the serial loop was scalar mask juggling); multiplying by lane
// constants keeps the chain pure VALU.
struct LaneK {
	int s1;      // stage-1 sign of own value: +1 for k < 2, -1 otherwise        (forward and inverse stage 2)
	int fa, fb;  // forward stage 2: out = fa * s + fb * q   -> (1,1) (-1,1) (1,2) (1,-2)
	int sh;      // inverse stage 1: own value >> sh, sh = 1 for k >= 2
	int ia, ib;  // inverse stage 1: e = ia * (d >> sh) + ib * q -> (1,1) (-1,1) (1,-1) (1,1)
};

DEVI LaneK lane_consts(int k)
{
	LaneK c;
	c.s1 = 1 - 2 * (k >> 1);
	c.fa = 1 - 2 * (int)(k == 1);
	c.fb = k < 2 ? 1 : (k == 2 ? 2 : -2);
	c.sh = k >> 1;
	c.ia = 1 - 2 * (int)(k == 1);
	c.ib = 1 - 2 * (int)(k == 2);
	return c;
}

// forward 4x4 core transform of the residual held one sample per lane (i = row,
// j = column). Returns W[P[i]][P[j]], P = {0,2,1,3}.
DEVI int fdct_lanes(int v, const LaneK &ci, const LaneK &cj)
{
	int p = dpp_xor3(v);
	int s = p + cj.s1 * v;                  // a b c d over j
	int q = dpp_xor1(s);
	s = cj.fa * s + cj.fb * q;              // W0 W2 W1 W3
	p = dpp_xor12(s);
	s = p + ci.s1 * s;
	q = dpp_xor4(s);
	return ci.fa * s + ci.fb * q;
}

// inverse of the above (rows first, then columns: clause 8.5.12.2); input is the
// scaled coefficient (P[i],P[j]) on lane (i,j); returns the residual sample (y=i, x=j)
DEVI int idct_lanes(int d, const LaneK &ci, const LaneK &cj)
{
	int q = dpp_xor1(d);
	int e = cj.ia * (d >> cj.sh) + cj.ib * q; // e0 e1 e2 e3
	int p = dpp_xor3(e);
	int f = p + cj.s1 * e;                    // f0 f1 f2 f3
	q = dpp_xor4(f);
	e = ci.ia * (f >> ci.sh) + ci.ib * q;
	p = dpp_xor12(e);
	f = p + ci.s1 * e;
	return (f + 32) >> 6;
}

// Every spin of the hand-off protocols is bounded in WALL time: s_memrealtime counts at 100 MHz
// whatever the shader clock does, so a wait is "lost" after WAIT_LIMIT_TICKS of real time and not
// after an iteration count that a busy or throttled GPU stretches (several sessions or processes
// polling on one device starve each other's producers for a while; that is slow, not lost). Looked
// at every 256th iteration; the first look starts the clock.
constexpr uint64_t WAIT_LIMIT_TICKS = 300000000ull; // 3 s
// The source planes of stream st (see Params::src_direct): base pointers; a reader takes row min(y, src_last_*) at stride cw.
DEVI const uint8_t *in_plane(const Params &p, int st, bool luma)
{
	if (p.in_luma)
		return luma ? p.in_luma[st] : p.in_chroma[st];
	const uint8_t *base = nullptr; // wave-uniform pick from the kernel arguments without indexing them dynamically
#pragma unroll
	for (int k = 0; k < IN_K_STREAMS; k++)
		base = st == k ? (luma ? p.in_k[k] : p.in_k[IN_K_STREAMS + k]) : base;
	return base;
}
DEVI const uint8_t *src_luma(const Params &p, int st) { return p.src_direct ? in_plane(p, st, true) : p.src_y + (size_t)st * p.plane_y_stride; }
DEVI const uint8_t *src_chroma(const Params &p, int st) { return p.src_direct ? in_plane(p, st, false) : p.src_uv + (size_t)st * p.plane_uv_stride; }

DEVI bool wait_expired(uint32_t spins, uint64_t &t0)
{
	if ((spins & 255u) != 255u)
		return false;
	const uint64_t now = __builtin_amdgcn_s_memrealtime();
	if (t0 == 0) {
		t0 = now | 1u;
		return false;
	}
	return now - t0 > WAIT_LIMIT_TICKS;
}

// development (CEDAR_DEBUG_HANG): who started / published / finished when, in 100 MHz ticks. One lane calls.
DEVI void dbg_event(const Params &p, uint32_t code, uint32_t a, uint32_t b)
{
	if (!p.dbg_log)
		return;
	const uint32_t i = atomicAdd(p.dbg_log, 1u);
	if (i >= (uint32_t)DBG_LOG_ENTRIES)
		return;
	const uint64_t t = __builtin_amdgcn_s_memrealtime();
	uint32_t *e = p.dbg_log + 4 + 4 * (size_t)i;
	e[0] = code | a << 8;
	e[1] = b;
	e[2] = (uint32_t)t;
	e[3] = (uint32_t)(t >> 32);
}

DEVI int mv_bits(int d)
{
	int a = iabs(d);
	return a == 0 ? 1 : 2 * (31 - __clz(8 * a)) + 1;
}

// regions of interest: QP_Y the macroblock is quantised with (the picture's + its region's offset)
// the picture's QP: one per launch, or -- rate control over several streams -- one per stream
DEVI int pic_qp(const Params &p, int st)
{
	return p.qp_s ? (int)p.qp_s[st] : p.qp;
}

DEVI int roi_qp(const Params &p, int st, int mx, int my)
{
	return min(max(pic_qp(p, st) + (int)p.roi_off[my * p.mb_w + mx], 1), 51);
}

// MAD classes: luma SAD / 256 of the macroblock against its chosen prediction -> one of five QP offsets
DEVI int mad_class_qp(const Params &p, int qp_base, int sad)
{
	const int mad = sad >> 8;
	const int cls = (int)(mad >= p.mad_th[0]) + (int)(mad >= p.mad_th[1]) + (int)(mad >= p.mad_th[2]) + (int)(mad >= p.mad_th[3]);
	const int dq = cls == 0 ? p.mad_dqp[0] : cls == 1 ? p.mad_dqp[1] : cls == 2 ? p.mad_dqp[2] : cls == 3 ? p.mad_dqp[3] : p.mad_dqp[4];
	return min(max(qp_base + dq, 1), 51);
}

// mb_qp_delta (7.4.5): against QP_Y of the macroblock before it in decoding order (k_qp_eff), the
// slice's at its start; 0 without regions of interest
DEVI int mb_qp_delta(const Params &p, int st, int mx, int my, int row0, size_t mbaddr)
{
	if (!p.roi_off)
		return 0;
	const int d = (int)p.qp_t[mbaddr] - ((mx == 0 && my == row0) ? pic_qp(p, st) : (int)p.qp_eff[mbaddr - 1]);
	// 7.4.5: the syntax element lies in -26..+25, QP_Y = (QP_Y,pred + mb_qp_delta + 52) % 52: a larger
	// step (two regions at +24 and -24 side by side) goes round the other way
	return d > 25 ? d - 52 : d < -26 ? d + 52 : d;
}

DEVI int median3(int a, int b, int c)
{
	int mn = a < b ? a : b, mx = a < b ? b : a;
	return c < mn ? mn : (c > mx ? mx : c);
}

} // namespace cedar
