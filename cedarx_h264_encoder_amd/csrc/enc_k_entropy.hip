// cedarx_h264_encoder_amd/csrc/enc_k_entropy.hip
// CAVLC back end on the GPU (clauses 7.3.5, 9.1, 9.2 and Annex B):
//   k_cavlc      every macroblock packs its own syntax elements into a
//                private bit buffer (nC contexts need only the neighbours'
//                coefficient counts, so this is fully parallel); P_Skip decided
//                here from the median motion-vector predictor
//   k_scan       per unit: mb_skip_run by a max-scan over skip flags, bit
//                offsets by a prefix sum of the bit lengths
//   k_zero/k_pack  bit-granular concatenation into the slice RBSP
//   k_esc_*      emulation prevention (00 00 0x -> 00 00 03 0x) as count / scan /
//                write over fixed chunks, NAL start code + header, compact output
// In the reference all of this is silicon behind PUTBITS/STM registers
// (kernel/cedar_regs.h:59-63; emulation prevention toggled at cedar.c:872-880).
#include <stdlib.h>

#include <hip/hip_runtime.h>

#include "enc_device.h"
#include "enc_tables.h"
#include "enc_wf_common.h" // (u32x4, the write-through loads and stores of the tagged hand-off words)

namespace cedar {

// ---------------- per-thread MSB-first bit writer into a slot ----------------
struct BitW {
	uint32_t *slot; // SLOT_WORDS words; each holds 32 stream bits, first bit = MSB
	uint64_t acc;
	int nacc;
	int words;
	int overflow;
};

DEVI void bw_put(BitW &b, uint32_t v, int n)
{
	if (n == 0)
		return;
	b.acc = (b.acc << n) | (uint64_t)(v & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u)));
	b.nacc += n;
	if (b.nacc >= 32) {
		uint32_t w = (uint32_t)(b.acc >> (b.nacc - 32));
		if (b.words < SLOT_WORDS)
			b.slot[b.words] = w;
		else
			b.overflow = 1;
		b.words++;
		b.nacc -= 32;
	}
}

DEVI void bw_ue(BitW &b, uint32_t v)
{
	v++;
	bw_put(b, v, 2 * (32 - __clz(v)) - 1);
}

DEVI void bw_se(BitW &b, int v)
{
	bw_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v));
}

DEVI int ue_len(uint32_t v)
{
	return 2 * (32 - __clz(v + 1)) - 1;
}

DEVI void bw_code(BitW &b, uint16_t packed)
{
	bw_put(b, packed & 0xff, packed >> 8);
}

DEVI void load_block(const int16_t *src, int lv[16])
{
	const u32x4 a = *(const u32x4 *)src, c = *(const u32x4 *)(src + 8);
	const uint32_t w[8] = { a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w };
#pragma unroll
	for (int i = 0; i < 8; i++) {
		lv[2 * i] = (int16_t)(w[i] & 0xffff);
		lv[2 * i + 1] = (int16_t)(w[i] >> 16);
	}
}

// ... of a macroblock whose levels are stored as bytes (MB_LEVELS_INT8): level index i at byte i
DEVI void load_block8(const int8_t *src, int lv[16])
{
	const u32x4 a = *(const u32x4 *)src;
	const uint32_t w[4] = { a.x, a.y, a.z, a.w };
#pragma unroll
	for (int i = 0; i < 16; i++)
		lv[i] = (int)(int8_t)((w[i >> 2] >> (8 * (i & 3))) & 0xffu);
}

// residual_block_cavlc (7.3.5.3.2 / 9.2): lv[first .. first+maxc) in scan order
DEVI void cavlc_block(BitW &b, const int lv[16], int first, int maxc, int nC)
{
	// positions of non-zero levels as a bitmask over scan index
	uint32_t mask = 0;
#pragma unroll
	for (int i = 0; i < 16; i++)
		if (i >= first && i < first + maxc && lv[i] != 0)
			mask |= 1u << (i - first);
	const int n = __popc(mask);
	int t1 = 0;
	{
		uint32_t m = mask;
		while (m && t1 < 3) {
			int hi = 31 - __clz(m);
			int v = 0;
#pragma unroll
			for (int i = 0; i < 16; i++)
				v = (i - first == hi) ? lv[i] : v;
			if (v != 1 && v != -1)
				break;
			t1++;
			m &= ~(1u << hi);
		}
	}
	if (nC < 0)
		bw_code(b, kCoeffTokenChromaDC[4 * n + t1]);
	else
		bw_code(b, kCoeffToken[(nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3) * 68 + 4 * n + t1]);
	if (n == 0)
		return;
	int suffix_len = (n > 10 && t1 < 3) ? 1 : 0;
	uint32_t m = mask;
	for (int k = 0; k < n; k++) {
		const int hi = 31 - __clz(m);
		m &= ~(1u << hi);
		int level = 0;
#pragma unroll
		for (int i = 0; i < 16; i++)
			level = (i - first == hi) ? lv[i] : level;
		if (k < t1) {
			bw_put(b, level < 0, 1);
			continue;
		}
		const int a = level < 0 ? -level : level;
		int code = 2 * a - 2 + (level < 0);
		if (k == t1 && t1 < 3)
			code -= 2;
		if (suffix_len == 0) {
			if (code < 14) {
				bw_put(b, 1, code + 1);
			} else if (code < 30) {
				bw_put(b, 1, 15);
				bw_put(b, (uint32_t)(code - 14), 4);
			} else {
				bw_put(b, 1, 16);
				bw_put(b, (uint32_t)(code - 30), 12);
			}
		} else if (code < (15 << suffix_len)) {
			bw_put(b, 1, (code >> suffix_len) + 1);
			bw_put(b, (uint32_t)(code & ((1 << suffix_len) - 1)), suffix_len);
		} else {
			bw_put(b, 1, 16);
			bw_put(b, (uint32_t)(code - (15 << suffix_len)), 12);
		}
		if (suffix_len == 0)
			suffix_len = 1;
		if (a > (3 << (suffix_len - 1)) && suffix_len < 6)
			suffix_len++;
	}
	const int last = 31 - __clz(mask);
	const int total_zeros = last + 1 - n;
	if (n < maxc) {
		if (maxc == 4)
			bw_code(b, kTotalZerosChromaDC[(n - 1) * 4 + total_zeros]);
		else
			bw_code(b, kTotalZeros[(n - 1) * 16 + total_zeros]);
	}
	int zeros_left = total_zeros;
	m = mask;
	for (int k = 0; k < n - 1 && zeros_left > 0; k++) {
		const int hi = 31 - __clz(m);
		m &= ~(1u << hi);
		const int next = 31 - __clz(m);
		const int run = hi - next - 1;
		bw_code(b, kRunBefore[((zeros_left > 7 ? 7 : zeros_left) - 1) * 16 + run]);
		zeros_left -= run;
	}
}

struct Nb {
	const uint8_t *cur, *left, *top; // nnz arrays (NULL = unavailable)
};

DEVI int calc_nc(const Nb &nb, int plane, int bx, int by)
{
	const int w = plane ? 2 : 4, base = plane ? 16 + 4 * (plane - 1) : 0;
	int na = -1, nbv = -1;
	if (bx > 0)
		na = nb.cur[base + bx - 1 + w * by];
	else if (nb.left)
		na = nb.left[base + w - 1 + w * by];
	if (by > 0)
		nbv = nb.cur[base + bx + w * (by - 1)];
	else if (nb.top)
		nbv = nb.top[base + bx + w * (w - 1)];
	if (na >= 0 && nbv >= 0)
		return (na + nbv + 1) >> 1;
	return na >= 0 ? na : nbv >= 0 ? nbv : 0;
}

DEVI void write_residual(BitW &b, const MbInfo &mi, const int16_t *cf, const Nb &nb)
{
	int lv[16];
	const bool c8 = mi.type == MB_P16x16 && (mi.chroma_mode & MB_LEVELS_INT8); // (k_inter: every level of the macroblock fits a byte)
	const int8_t *cf8 = (const int8_t *)cf;
	if (mi.type == MB_I16x16) {
		load_block(cf + COEF_DC16, lv);
		cavlc_block(b, lv, 0, 16, calc_nc(nb, 0, 0, 0));
	}
	for (int i8 = 0; i8 < 4; i8++) {
		if (!(mi.cbp & (1 << i8)))
			continue;
		for (int i4 = 0; i4 < 4; i4++) {
			const int idx = 4 * i8 + i4;
			const int nC = calc_nc(nb, 0, blk_x(idx), blk_y(idx));
			if (c8)
				load_block8(cf8 + COEF_LUMA + idx * 16, lv);
			else
				load_block(cf + COEF_LUMA + idx * 16, lv);
			if (mi.type == MB_I16x16)
				cavlc_block(b, lv, 1, 15, nC);
			else
				cavlc_block(b, lv, 0, 16, nC);
		}
	}
	if (mi.cbp & 0x30) {
		uint32_t w[4];
		if (c8) { // 8 bytes at index COEF_CDC
			const uint64_t d = *(const uint64_t *)(cf8 + COEF_CDC);
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t lo = (uint32_t)(d >> (16 * k)) & 0xffu, hi = (uint32_t)(d >> (16 * k + 8)) & 0xffu;
				w[k] = ((uint32_t)(int)(int8_t)lo & 0xffffu) | ((uint32_t)(int)(int8_t)hi << 16);
			}
		} else {
			const u32x4 d = *(const u32x4 *)(cf + COEF_CDC);
			w[0] = d.x; w[1] = d.y; w[2] = d.z; w[3] = d.w;
		}
		for (int c = 0; c < 2; c++) {
#pragma unroll
			for (int i = 0; i < 16; i++)
				lv[i] = 0;
			lv[0] = (int16_t)(w[2 * c] & 0xffff);
			lv[1] = (int16_t)(w[2 * c] >> 16);
			lv[2] = (int16_t)(w[2 * c + 1] & 0xffff);
			lv[3] = (int16_t)(w[2 * c + 1] >> 16);
			cavlc_block(b, lv, 0, 4, -1);
		}
		if (mi.cbp & 0x20)
			for (int c = 0; c < 2; c++)
				for (int blk = 0; blk < 4; blk++) {
					if (c8) { // (16-byte loads want 16-byte addresses: index 280 + 16 k is 8 modulo 16)
						const int8_t *q = cf8 + COEF_CAC + (4 * c + blk) * 16;
						const uint64_t d0 = *(const uint64_t *)q, d1 = *(const uint64_t *)(q + 8);
#pragma unroll
						for (int i = 0; i < 8; i++) {
							lv[i] = (int)(int8_t)((d0 >> (8 * i)) & 0xffu);
							lv[8 + i] = (int)(int8_t)((d1 >> (8 * i)) & 0xffu);
						}
					} else {
						load_block(cf + COEF_CAC + (4 * c + blk) * 16, lv);
					}
					cavlc_block(b, lv, 1, 15, calc_nc(nb, 1 + c, blk & 1, blk >> 1));
				}
	}
}

// ------------------------------------------------------------------------
// k_cavlc: one thread per macroblock
// ------------------------------------------------------------------------
__global__ __launch_bounds__(64) void k_cavlc(Params p)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x; // MB index within the encoded rows
	const int st = blockIdx.y;
	if (i >= p.row_num * p.mb_w)
		return;
	const int my = p.row_first + i / p.mb_w, mx = i % p.mb_w;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, slice_of_row(p.mb_h, p.slice_count, my), &r0, &r1);
	const size_t mbaddr = (size_t)st * p.n_mb + (size_t)my * p.mb_w + mx;
	const bool has_left = mx > 0, has_top = my > r0;

	MbInfo mi = p.mbi[mbaddr];
	BitW b;
	b.slot = p.slot + mbaddr * SLOT_WORDS;
	b.acc = 0;
	b.nacc = 0;
	b.words = 0;
	b.overflow = 0;

	Nb nb;
	nb.cur = p.nnz + mbaddr * NNZ_PER_MB;
	nb.left = has_left ? p.nnz + (mbaddr - 1) * NNZ_PER_MB : nullptr;
	nb.top = has_top ? p.nnz + (mbaddr - p.mb_w) * NNZ_PER_MB : nullptr;
	const int16_t *cf = p.coef + mbaddr * COEF_PER_MB;
	int skip = 0;

	const bool is_pcm = mi.type == MB_IPCM;
	if (is_pcm) {
		// I_PCM (7.3.5): mb_type 25 (+ 5 in a P slice); pcm_alignment_zero_bits and the 384 samples are placed by
		// k_pack, straight from the source planes, at the byte position k_scan finds
		bw_ue(b, (uint32_t)(p.is_idr ? 25 : 30));
	} else if (p.is_idr) {
		if (mi.type == MB_I4x4) {
			bw_ue(b, 0);
			const int8_t *mo = p.i4modes + mbaddr * 16;
			for (int idx = 0; idx < 16; idx++) {
				const int bx = blk_x(idx), by = blk_y(idx);
				// predicted mode (8.3.1.1): DC when either neighbour MB is unavailable
				int ma = 2, mb_ = 2;
				bool dcflag = false;
				if (bx > 0) {
					ma = mo[idx - (bx & 1 ? 1 : 3)]; // left block inside the MB
				} else if (has_left) {
					const MbInfo ml = p.mbi[mbaddr - 1];
					ma = ml.type == MB_I4x4 ? p.i4modes[(mbaddr - 1) * 16 + idx + 5] : 2;
				} else {
					dcflag = true;
				}
				if (by > 0) {
					mb_ = mo[idx - (by & 1 ? 2 : 6)]; // block above inside the MB
				} else if (has_top) {
					const MbInfo mt = p.mbi[mbaddr - p.mb_w];
					mb_ = mt.type == MB_I4x4 ? p.i4modes[(mbaddr - p.mb_w) * 16 + idx + 10] : 2;
				} else {
					dcflag = true;
				}
				const int pm = dcflag ? 2 : (ma < mb_ ? ma : mb_);
				const int m = mo[idx];
				if (m == pm) {
					bw_put(b, 1, 1);
				} else {
					bw_put(b, 0, 1);
					bw_put(b, (uint32_t)(m < pm ? m : m - 1), 3);
				}
			}
			bw_ue(b, (uint32_t)mi.chroma_mode);
			bw_ue(b, kCbpCodeIntra[mi.cbp & 0x3f]);
			if (mi.cbp & 0x3f)
				bw_se(b, mb_qp_delta(p, st, mx, my, r0, mbaddr));
		} else {
			const int t = 1 + mi.i16_mode + 4 * ((mi.cbp >> 4) & 3) + ((mi.cbp & 15) ? 12 : 0);
			bw_ue(b, (uint32_t)t);
			bw_ue(b, (uint32_t)mi.chroma_mode);
			bw_se(b, mb_qp_delta(p, st, mx, my, r0, mbaddr));
		}
		write_residual(b, mi, cf, nb);
	} else if (mi.type == MB_I16x16) {
		// a refresh macroblock (Params::intra_refresh): intra in a P slice, mb_type 5 + the I slice's
		const int t = 1 + mi.i16_mode + 4 * ((mi.cbp >> 4) & 3) + ((mi.cbp & 15) ? 12 : 0);
		bw_ue(b, (uint32_t)(5 + t));
		bw_ue(b, (uint32_t)mi.chroma_mode);
		bw_se(b, mb_qp_delta(p, st, mx, my, r0, mbaddr));
		write_residual(b, mi, cf, nb);
	} else {
		// motion vector prediction (8.4.1.3) and P_Skip inference (8.4.1.1). An intra neighbour (a
		// refresh macroblock) is available with refIdx -1 and a zero vector; ra/rb/rc = "has refIdx 0".
		int ax = 0, ay = 0, bx_ = 0, by_ = 0, cx = 0, cy = 0;
		bool hasA = has_left, hasB = has_top, hasC = false;
		bool ra = false, rb = false, rc = false;
		if (hasA) {
			const MbInfo m = p.mbi[mbaddr - 1];
			ra = m.type >= MB_P16x16;
			ax = m.mvx; ay = m.mvy; // (zero in an intra record)
		}
		if (hasB) {
			const MbInfo m = p.mbi[mbaddr - p.mb_w];
			rb = m.type >= MB_P16x16;
			bx_ = m.mvx; by_ = m.mvy;
			if (mx < p.mb_w - 1) {
				const MbInfo c = p.mbi[mbaddr - p.mb_w + 1];
				rc = c.type >= MB_P16x16;
				cx = c.mvx; cy = c.mvy; hasC = true;
			} else if (mx > 0) {
				const MbInfo c = p.mbi[mbaddr - p.mb_w - 1];
				rc = c.type >= MB_P16x16;
				cx = c.mvx; cy = c.mvy; hasC = true;
			}
		}
		int px, py;
		const int n = (int)ra + (int)rb + (int)rc;
		if (!hasB && !hasC && hasA) {
			px = ax; py = ay;
		} else if (n == 1) {
			px = ra ? ax : rb ? bx_ : cx;
			py = ra ? ay : rb ? by_ : cy;
		} else {
			px = median3(ax, bx_, cx);
			py = median3(ay, by_, cy);
		}
		int skx = px, sky = py;
		if (!hasA || !hasB || (ra && ax == 0 && ay == 0) || (rb && bx_ == 0 && by_ == 0)) {
			skx = 0; sky = 0;
		}
		if (mi.cbp == 0 && mi.mvx == skx && mi.mvy == sky) {
			skip = 1;
			mi.type = MB_PSKIP;
			p.mbi[mbaddr].type = MB_PSKIP; // nobody else reads .type of a P macroblock concurrently
		} else {
			bw_ue(b, 0); // P_L0_16x16
			bw_se(b, mi.mvx - px); // vectors are kept in quarter samples
			bw_se(b, mi.mvy - py);
			bw_ue(b, kCbpCodeInter[mi.cbp & 0x3f]);
			if (mi.cbp & 0x3f)
				bw_se(b, mb_qp_delta(p, st, mx, my, r0, mbaddr));
			write_residual(b, mi, cf, nb);
		}
	}
	const uint32_t total = (uint32_t)(b.words * 32 + b.nacc);
	if (b.nacc > 0) {
		if (b.words < SLOT_WORDS)
			b.slot[b.words] = (uint32_t)(b.acc << (32 - b.nacc));
		else
			b.overflow = 1;
	}
	p.mb_bits[mbaddr] = is_pcm ? total | PCM_MB_BITS_FLAG : total;
	p.skip[mbaddr] = (uint8_t)skip;
	if (b.overflow)
		atomicOr(&p.stream_size[p.n_streams + 1], 1u);
	if (total > MB_BITS_LIMIT) // H.264 A.3.1; the session counts or refuses the picture
		atomicOr(&p.stream_size[p.n_streams + 1], 4u);
}

// ------------------------------------------------------------------------
// k_cavlc2: one wave codes TWO macroblocks, one lane per syntax part:
//   lane 0 of a half   macroblock header (type, prediction modes / mvd, cbp, qp delta;
//                      P_Skip inference)
//   lane 1             Intra16x16 DC block
//   lanes 2..17        the 16 luma 4x4 blocks (coding order)
//   lanes 18, 19       chroma DC Cb, Cr
//   lanes 20..27       chroma AC Cb 0..3, Cr 0..3
// Every lane runs residual_block_cavlc on its block into a private LDS bit buffer;
// a prefix sum of the lengths gives each part its bit position, the parts are ORed
// into the macroblock's slot image in LDS and stored with coalesced writes.
// Thread-per-macroblock (the previous form) left a 1080p picture with 128 waves of
// serial work, 0.25 ms; this form has 4080 waves of 1/28 the chain.
// ------------------------------------------------------------------------
constexpr int CV_LANE_WORDS = 24; // 768 bits >= the longest coded 4x4 block (16 escapes + token + zeros + runs < 640)

struct LdsBitW {
	uint32_t *buf;
	uint64_t acc;
	int nacc, words, overflow;
};

DEVI void lw_put(LdsBitW &b, uint32_t v, int n)
{
	if (n == 0)
		return;
	b.acc = (b.acc << n) | (uint64_t)(v & (n >= 32 ? 0xffffffffu : ((1u << n) - 1u)));
	b.nacc += n;
	if (b.nacc >= 32) {
		if (b.words < CV_LANE_WORDS)
			b.buf[b.words] = (uint32_t)(b.acc >> (b.nacc - 32));
		else
			b.overflow = 1;
		b.words++;
		b.nacc -= 32;
	}
}
DEVI void lw_ue(LdsBitW &b, uint32_t v)
{
	v++;
	lw_put(b, v, 2 * (32 - __clz(v)) - 1);
}
DEVI void lw_se(LdsBitW &b, int v) { lw_ue(b, v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); }
DEVI void lw_code(LdsBitW &b, uint16_t packed) { lw_put(b, packed & 0xff, packed >> 8); }

// residual_block_cavlc (7.3.5.3.2 / 9.2) of lv[first .. first+maxc), levels read from LDS
DEVI void cavlc_block_lds(LdsBitW &b, const int16_t *lv, int first, int maxc, int nC)
{
	uint32_t mask = 0;
	for (int i = 0; i < maxc; i++)
		mask |= lv[first + i] != 0 ? 1u << i : 0u;
	const int n = __popc(mask);
	int t1 = 0;
	for (uint32_t m = mask; m && t1 < 3; t1++) {
		const int hi = 31 - __clz(m);
		const int v = lv[first + hi];
		if (v != 1 && v != -1)
			break;
		m &= ~(1u << hi);
	}
	if (nC < 0)
		lw_code(b, kCoeffTokenChromaDC[4 * n + t1]);
	else
		lw_code(b, kCoeffToken[(nC < 2 ? 0 : nC < 4 ? 1 : nC < 8 ? 2 : 3) * 68 + 4 * n + t1]);
	if (n == 0)
		return;
	int suffix_len = (n > 10 && t1 < 3) ? 1 : 0;
	uint32_t m = mask;
	for (int k = 0; k < n; k++) {
		const int hi = 31 - __clz(m);
		m &= ~(1u << hi);
		const int level = lv[first + hi];
		if (k < t1) {
			lw_put(b, level < 0, 1);
			continue;
		}
		const int a = level < 0 ? -level : level;
		int code = 2 * a - 2 + (level < 0);
		if (k == t1 && t1 < 3)
			code -= 2;
		if (suffix_len == 0) {
			if (code < 14) {
				lw_put(b, 1, code + 1);
			} else if (code < 30) {
				lw_put(b, 1, 15);
				lw_put(b, (uint32_t)(code - 14), 4);
			} else {
				lw_put(b, 1, 16);
				lw_put(b, (uint32_t)(code - 30), 12);
			}
		} else if (code < (15 << suffix_len)) {
			lw_put(b, 1, (code >> suffix_len) + 1);
			lw_put(b, (uint32_t)(code & ((1 << suffix_len) - 1)), suffix_len);
		} else {
			lw_put(b, 1, 16);
			lw_put(b, (uint32_t)(code - (15 << suffix_len)), 12);
		}
		if (suffix_len == 0)
			suffix_len = 1;
		if (a > (3 << (suffix_len - 1)) && suffix_len < 6)
			suffix_len++;
	}
	const int last = 31 - __clz(mask);
	const int total_zeros = last + 1 - n;
	if (n < maxc) {
		if (maxc == 4)
			lw_code(b, kTotalZerosChromaDC[(n - 1) * 4 + total_zeros]);
		else
			lw_code(b, kTotalZeros[(n - 1) * 16 + total_zeros]);
	}
	int zeros_left = total_zeros;
	m = mask;
	for (int k = 0; k < n - 1 && zeros_left > 0; k++) {
		const int hi = 31 - __clz(m);
		m &= ~(1u << hi);
		const int next = 31 - __clz(m);
		const int run = hi - next - 1;
		lw_code(b, kRunBefore[((zeros_left > 7 ? 7 : zeros_left) - 1) * 16 + run]);
		zeros_left -= run;
	}
}

__global__ __launch_bounds__(64) void k_cavlc2(Params p)
{
	__shared__ __attribute__((aligned(16))) int16_t s_cf[2][COEF_PER_MB];
	__shared__ __attribute__((aligned(16))) uint8_t s_nnz[2][3][32]; // cur, left, top
	__shared__ __attribute__((aligned(16))) int8_t s_modes[2][3][16];
	__shared__ MbInfo s_mi[2][4];                                      // cur, left, top, top-right / top-left
	__shared__ uint32_t s_lane[64][CV_LANE_WORDS];
	__shared__ uint32_t s_out[2][SLOT_WORDS];

	const int lane = threadIdx.x, half = lane >> 5, hl = lane & 31;
	const int i = blockIdx.x * 2 + half; // macroblock index within the encoded rows
	const int st = blockIdx.y;
	const int nmb = p.row_num * p.mb_w;
	const bool valid = i < nmb;
	const int ci = valid ? i : nmb - 1;
	const int my = p.row_first + ci / p.mb_w, mx = ci % p.mb_w;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, slice_of_row(p.mb_h, p.slice_count, my), &r0, &r1);
	const size_t mbaddr = (size_t)st * p.n_mb + (size_t)my * p.mb_w + mx;
	const bool has_left = mx > 0, has_top = my > r0;
	const bool has_c = has_top && (mx < p.mb_w - 1 || mx > 0);
	// (has_c, not has_top: a picture one macroblock wide has no neighbour C at all, and mbaddr - 2 of
	// its second row is element -1 of the record array)
	const size_t addr_c = has_c ? (mx < p.mb_w - 1 ? mbaddr - p.mb_w + 1 : mbaddr - p.mb_w - 1) : mbaddr;

	// ---- stage everything the half needs (coalesced) ----
	{
		// levels: only the block groups the macroblock's coded block pattern names (k_inter stores no others;
		// 16-byte chunk hl = half a luma block, hl + 32 = Intra16x16 DCs, chroma DCs, chroma AC halves) -- or, where k_inter
		// stored them as bytes (MB_LEVELS_INT8), chunk hl = the 16 levels from index 16 hl on, widened on the way into LDS
		const MbInfo m0 = p.mbi[mbaddr];
		const bool i16 = m0.type == MB_I16x16;
		const u32x4 *src = (const u32x4 *)(p.coef + mbaddr * COEF_PER_MB);
		u32x4 *dst = (u32x4 *)s_cf[half];
		if (m0.type == MB_P16x16 && (m0.chroma_mode & MB_LEVELS_INT8)) {
			const bool need = hl < 16 ? ((m0.cbp >> (hl >> 2)) & 1) != 0 : hl == 17 ? (m0.cbp & 0x30) != 0 : hl >= 18 && hl < 26 && (m0.cbp & 0x20) != 0;
			if (need) {
				const u32x4 v = src[hl];
				auto wide = [](uint32_t w, int k) { // bytes 2k, 2k + 1 of w as two int16
					const uint32_t lo = (w >> (16 * k)) & 0xffu, hi = (w >> (16 * k + 8)) & 0xffu;
					return ((uint32_t)(int)(int8_t)lo & 0xffffu) | ((uint32_t)(int)(int8_t)hi << 16);
				};
				dst[2 * hl] = u32x4{ wide(v.x, 0), wide(v.x, 1), wide(v.y, 0), wide(v.y, 1) };
				if (hl < 25) // (levels 400 .. 407 are the last)
					dst[2 * hl + 1] = u32x4{ wide(v.z, 0), wide(v.z, 1), wide(v.w, 0), wide(v.w, 1) };
			}
		} else {
			if ((m0.cbp >> (hl >> 3)) & 1)
				dst[hl] = src[hl];
			const int c2 = hl + 32;
			if (c2 < COEF_PER_MB * 2 / 16 && (c2 < 34 ? i16 : c2 == 34 ? (m0.cbp & 0x30) != 0 : (m0.cbp & 0x20) != 0))
				dst[c2] = src[c2];
		}
		if (hl < 24) {
			s_nnz[half][0][hl] = p.nnz[mbaddr * NNZ_PER_MB + hl];
			s_nnz[half][1][hl] = has_left ? p.nnz[(mbaddr - 1) * NNZ_PER_MB + hl] : 0;
			s_nnz[half][2][hl] = has_top ? p.nnz[(mbaddr - p.mb_w) * NNZ_PER_MB + hl] : 0;
		}
		if (hl < 16 && p.is_idr) {
			s_modes[half][0][hl] = p.i4modes[mbaddr * 16 + hl];
			s_modes[half][1][hl] = has_left ? p.i4modes[(mbaddr - 1) * 16 + hl] : 2;
			s_modes[half][2][hl] = has_top ? p.i4modes[(mbaddr - p.mb_w) * 16 + hl] : 2;
		}
		if (hl < 4) {
			const size_t a = hl == 0 ? mbaddr : hl == 1 ? (has_left ? mbaddr - 1 : mbaddr)
					 : hl == 2 ? (has_top ? mbaddr - p.mb_w : mbaddr) : addr_c;
			s_mi[half][hl] = p.mbi[a];
		}
	}
	__syncthreads();

	const MbInfo mi = s_mi[half][0];
	const int16_t *cf = s_cf[half];
	Nb nb;
	nb.cur = s_nnz[half][0];
	nb.left = has_left ? s_nnz[half][1] : nullptr;
	nb.top = has_top ? s_nnz[half][2] : nullptr;
	LdsBitW b;
	b.buf = s_lane[lane];
	b.acc = 0;
	b.nacc = b.words = b.overflow = 0;
	int skip = 0;
	const int cbp = mi.cbp & 0x3f;

	const bool is_pcm = mi.type == MB_IPCM;
	if (valid && hl == 0) {
		// ---- macroblock header ----
		if (is_pcm) { // I_PCM: mb_type only; alignment and samples come from k_pack (see k_cavlc)
			lw_ue(b, (uint32_t)(p.is_idr ? 25 : 30));
		} else if (p.is_idr) {
			if (mi.type == MB_I4x4) {
				lw_ue(b, 0);
				const int8_t *mo = s_modes[half][0];
				const bool left_i4 = has_left && s_mi[half][1].type == MB_I4x4;
				const bool top_i4 = has_top && s_mi[half][2].type == MB_I4x4;
				for (int idx = 0; idx < 16; idx++) {
					const int bx = blk_x(idx), by = blk_y(idx);
					// predicted mode (8.3.1.1): DC when either neighbour MB is unavailable
					int ma = 2, mb_ = 2;
					bool dcflag = false;
					if (bx > 0)
						ma = mo[idx - (bx & 1 ? 1 : 3)];
					else if (has_left)
						ma = left_i4 ? s_modes[half][1][idx + 5] : 2;
					else
						dcflag = true;
					if (by > 0)
						mb_ = mo[idx - (by & 1 ? 2 : 6)];
					else if (has_top)
						mb_ = top_i4 ? s_modes[half][2][idx + 10] : 2;
					else
						dcflag = true;
					const int pm = dcflag ? 2 : (ma < mb_ ? ma : mb_);
					const int m = mo[idx];
					if (m == pm) {
						lw_put(b, 1, 1);
					} else {
						lw_put(b, 0, 1);
						lw_put(b, (uint32_t)(m < pm ? m : m - 1), 3);
					}
				}
				lw_ue(b, (uint32_t)mi.chroma_mode);
				lw_ue(b, kCbpCodeIntra[cbp]);
				if (cbp)
					lw_se(b, mb_qp_delta(p, st, mx, my, r0, mbaddr));
			} else {
				const int t = 1 + mi.i16_mode + 4 * ((mi.cbp >> 4) & 3) + ((mi.cbp & 15) ? 12 : 0);
				lw_ue(b, (uint32_t)t);
				lw_ue(b, (uint32_t)mi.chroma_mode);
				lw_se(b, mb_qp_delta(p, st, mx, my, r0, mbaddr));
			}
		} else if (mi.type == MB_I16x16) {
			// a refresh macroblock (Params::intra_refresh): intra in a P slice, mb_type 5 + the I slice's
			const int t = 1 + mi.i16_mode + 4 * ((mi.cbp >> 4) & 3) + ((mi.cbp & 15) ? 12 : 0);
			lw_ue(b, (uint32_t)(5 + t));
			lw_ue(b, (uint32_t)mi.chroma_mode);
			lw_se(b, mb_qp_delta(p, st, mx, my, r0, mbaddr));
		} else {
			// motion vector prediction (8.4.1.3) and P_Skip inference (8.4.1.1). An intra neighbour (a
			// refresh macroblock) is available with refIdx -1 and a zero vector; ra/rb/rc = "has refIdx 0".
			const bool hasA = has_left, hasB = has_top, hasC = has_c;
			const MbInfo mA = s_mi[half][1], mB = s_mi[half][2], mC = s_mi[half][3];
			const bool ra = hasA && mA.type >= MB_P16x16, rb = hasB && mB.type >= MB_P16x16, rc = hasC && mC.type >= MB_P16x16;
			const int ax = ra ? mA.mvx : 0, ay = ra ? mA.mvy : 0;
			const int bx_ = rb ? mB.mvx : 0, by_ = rb ? mB.mvy : 0;
			const int cx = rc ? mC.mvx : 0, cy = rc ? mC.mvy : 0;
			int px, py;
			const int n = (int)ra + (int)rb + (int)rc;
			if (!hasB && !hasC && hasA) {
				px = ax; py = ay;
			} else if (n == 1) {
				px = ra ? ax : rb ? bx_ : cx;
				py = ra ? ay : rb ? by_ : cy;
			} else {
				px = median3(ax, bx_, cx);
				py = median3(ay, by_, cy);
			}
			int skx = px, sky = py;
			if (!hasA || !hasB || (ra && ax == 0 && ay == 0) || (rb && bx_ == 0 && by_ == 0)) {
				skx = 0; sky = 0;
			}
			if (mi.cbp == 0 && mi.mvx == skx && mi.mvy == sky) {
				skip = 1;
				p.mbi[mbaddr].type = MB_PSKIP; // nobody else reads .type of a P macroblock concurrently
			} else {
				lw_ue(b, 0); // P_L0_16x16
				lw_se(b, mi.mvx - px); // vectors are kept in quarter samples
				lw_se(b, mi.mvy - py);
				lw_ue(b, kCbpCodeInter[cbp]);
				if (cbp)
					lw_se(b, mb_qp_delta(p, st, mx, my, r0, mbaddr));
			}
		}
	} else if (valid && hl == 1) {
		if (mi.type == MB_I16x16)
			cavlc_block_lds(b, cf + COEF_DC16, 0, 16, calc_nc(nb, 0, 0, 0));
	} else if (valid && hl < 18) {
		const int idx = hl - 2;
		if (cbp & (1 << (idx >> 2))) {
			const int nC = calc_nc(nb, 0, blk_x(idx), blk_y(idx));
			if (mi.type == MB_I16x16)
				cavlc_block_lds(b, cf + COEF_LUMA + idx * 16, 1, 15, nC);
			else
				cavlc_block_lds(b, cf + COEF_LUMA + idx * 16, 0, 16, nC);
		}
	} else if (valid && hl < 20) {
		if (cbp & 0x30)
			cavlc_block_lds(b, cf + COEF_CDC + 4 * (hl - 18), 0, 4, -1);
	} else if (valid && hl < 28) {
		if (cbp & 0x20) {
			const int c = (hl - 20) >> 2, blk = (hl - 20) & 3;
			cavlc_block_lds(b, cf + COEF_CAC + (4 * c + blk) * 16, 1, 15, calc_nc(nb, 1 + c, blk & 1, blk >> 1));
		}
	}
	// flush the partial word, left-aligned
	if (b.nacc > 0) {
		if (b.words < CV_LANE_WORDS)
			b.buf[b.words] = (uint32_t)(b.acc << (32 - b.nacc));
		else
			b.overflow = 1;
	}
	const int my_bits = b.words * 32 + b.nacc;

	// ---- bit position of every part: exclusive prefix over the 32 lanes of the half ----
	int incl = my_bits;
#pragma unroll
	for (int off = 1; off < 32; off <<= 1) {
		const int o = __shfl_up(incl, off, 32);
		if (hl >= off)
			incl += o;
	}
	const int total = __shfl(incl, 31, 32);
	const int nwords = (total + 31) >> 5;
	for (int w = hl; w < nwords && w < SLOT_WORDS; w += 32)
		s_out[half][w] = 0;
	__syncthreads();
	{
		uint32_t pos = (uint32_t)(incl - my_bits);
		const int nw = b.words + (b.nacc > 0 ? 1 : 0);
		for (int w = 0; w < nw && w < CV_LANE_WORDS; w++) {
			const int nbt = w < b.words ? 32 : b.nacc;
			const uint32_t v = b.buf[w] >> (32 - nbt); // right-aligned
			const uint32_t wi = pos >> 5, sft = pos & 31;
			if (wi + 1 < (uint32_t)SLOT_WORDS) {
				if (sft + nbt <= 32) {
					atomicOr(&s_out[half][wi], v << (32 - sft - nbt));
				} else {
					const int lo = (int)sft + nbt - 32;
					atomicOr(&s_out[half][wi], v >> lo);
					atomicOr(&s_out[half][wi + 1], v << (32 - lo));
				}
			}
			pos += nbt;
		}
	}
	__syncthreads();
	if (valid) {
		uint32_t *slot = p.slot + mbaddr * SLOT_WORDS;
		for (int w = hl; w < nwords && w < SLOT_WORDS; w += 32)
			slot[w] = s_out[half][w];
		if (hl == 0) {
			p.mb_bits[mbaddr] = is_pcm ? (uint32_t)total | PCM_MB_BITS_FLAG : (uint32_t)total;
			p.skip[mbaddr] = (uint8_t)skip;
		}
		if (b.overflow || (hl == 0 && nwords + 1 >= SLOT_WORDS))
			atomicOr(&p.stream_size[p.n_streams + 1], 1u);
		if (hl == 0 && total > MB_BITS_LIMIT)
			atomicOr(&p.stream_size[p.n_streams + 1], 4u);
	}
}

// ---------------- block-wide inclusive scans (1024 threads) ----------------
template <bool IS_MAX>
DEVI int block_scan_incl(int v, int *s_tmp)
{
	// (any block size up to 1024 threads: the waves that do not exist contribute the identity)
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (int)(blockDim.x >> 6);
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		int o = __shfl_up(v, off, 64);
		if (lane >= off)
			v = IS_MAX ? (o > v ? o : v) : v + o;
	}
	__syncthreads();
	if (lane == 63)
		s_tmp[wave] = v;
	__syncthreads();
	if (wave == 0) {
		int t = lane < nwaves ? s_tmp[lane] : (IS_MAX ? -0x7fffffff : 0);
#pragma unroll
		for (int off = 1; off < 16; off <<= 1) {
			int o = __shfl_up(t, off, 64);
			if (lane >= off)
				t = IS_MAX ? (o > t ? o : t) : t + o;
		}
		if (lane < 16)
			s_tmp[16 + lane] = t;
	}
	__syncthreads();
	if (wave > 0) {
		int pre = s_tmp[16 + wave - 1];
		v = IS_MAX ? (pre > v ? pre : v) : v + pre;
	}
	return v;
}

// Positions in a slice with I_PCM macroblocks (k_scan). f(pos) = pcm ? align8(pos + a) + b : pos + a.
struct PosMap {
	int a, b, pcm;
};

DEVI int posmap_apply(const PosMap &f, int pos)
{
	return f.pcm ? ((pos + f.a + 7) & ~7) + f.b : pos + f.a;
}

// g after f. align8(x + y) = x + align8(y) for x a multiple of 8: behind an I_PCM macroblock everything is aligned.
DEVI PosMap posmap_then(const PosMap &f, const PosMap &g)
{
	if (!g.pcm)
		return f.pcm ? PosMap{ f.a, f.b + g.a, 1 } : PosMap{ f.a + g.a, 0, 0 };
	return f.pcm ? PosMap{ f.a, ((f.b + g.a + 7) & ~7) + g.b, 1 } : PosMap{ f.a + g.a, g.b, 1 };
}

// inclusive scan of position maps over the block (the structure of block_scan_incl)
DEVI PosMap block_scan_posmap(PosMap v, PosMap *s_pm)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (int)(blockDim.x >> 6);
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const PosMap o{ __shfl_up(v.a, off, 64), __shfl_up(v.b, off, 64), __shfl_up(v.pcm, off, 64) };
		if (lane >= off)
			v = posmap_then(o, v);
	}
	__syncthreads();
	if (lane == 63)
		s_pm[wave] = v;
	__syncthreads();
	if (wave == 0) {
		PosMap t = lane < nwaves ? s_pm[lane] : PosMap{ 0, 0, 0 };
#pragma unroll
		for (int off = 1; off < 16; off <<= 1) {
			const PosMap o{ __shfl_up(t.a, off, 64), __shfl_up(t.b, off, 64), __shfl_up(t.pcm, off, 64) };
			if (lane >= off)
				t = posmap_then(o, t);
		}
		if (lane < 16)
			s_pm[16 + lane] = t;
	}
	__syncthreads();
	if (wave > 0)
		v = posmap_then(s_pm[16 + wave - 1], v);
	return v;
}

// ------------------------------------------------------------------------
// k_scan: one SCAN_THREADS-thread block per unit. Entry e of bit_off / run_before:
//   e in [0, n): macroblock e of the unit; e == n: slice trailer.
// ------------------------------------------------------------------------
// The two per-unit scan kernels take their workgroup size from the launch: 1024 threads for a few
// units (fewest loop trips: the single stream's entropy chain is latency), 256 for a batch. 1024
// threads are 16 waves, which only an idle CU can take: beside a batch's k_inter, whose small
// workgroups refill every CU as fast as they drain, a launch of 32 such blocks waited ~0.6 ms for
// room to do 20 us of work (profiles/r02_s32: k_scan 675 us average). Four waves slip in anywhere.
#define SCAN_THREADS ((int)blockDim.x)

// (round 4: a thread takes SC_PER_T consecutive entries -- a serial scan of eight in registers around ONE block scan per
// tile of 8 x threads entries: a 1080p unit is a single tile for 1024 threads where it was 8 trips of 4 barriers each,
// 25 -> ~8 us alone; a batch's 256-thread workgroups make 4 trips instead of 32)
constexpr int SC_PER_T = 8;

__global__ __launch_bounds__(1024) void k_scan(Params p)
{
	__shared__ int s_tmp[32];
	__shared__ PosMap s_pm[32];
	__shared__ int s_carry_max, s_carry_sum;
	const int unit = blockIdx.x;
	const int st = unit / p.slice_num, ls = unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, p.slice_first + ls, &r0, &r1);
	const int n = (r1 - r0) * p.mb_w;
	const size_t mb0 = (size_t)st * p.n_mb + (size_t)r0 * p.mb_w;
	uint32_t *off = p.bit_off + (size_t)unit * (p.unit_mb_cap + 2);
	uint32_t *run = p.run_before + (size_t)unit * (p.unit_mb_cap + 2);
	const bool is_p = !p.is_idr;
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (int)(blockDim.x >> 6);
	if (threadIdx.x == 0) {
		s_carry_max = -1;
		s_carry_sum = p.hdr[unit].nbits;
		// k_pack_out's ticket counter starts at zero (it cannot clear the word itself: a workgroup that starts after the
		// launch's last chunk is done must still find a ticket that says so)
		if (p.ent_rec && unit == 0)
			__hip_atomic_store(&p.stream_size[p.n_streams + 3], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	}
	__syncthreads();
	for (int base = 0; base <= n; base += SCAN_THREADS * SC_PER_T) {
		const int e0 = base + (int)threadIdx.x * SC_PER_T;
		// ---- this thread's entries: skip flags and bits ----
		uint32_t mbb[SC_PER_T];
		bool coded[SC_PER_T];
#pragma unroll
		for (int i = 0; i < SC_PER_T; i++) {
			const int e = e0 + i;
			const bool mb = e < n;
			coded[i] = mb && !p.skip[mb0 + (mb ? e : 0)];
			mbb[i] = coded[i] ? p.mb_bits[mb0 + e] : 0u;
		}
		// ---- last coded macroblock at or before every entry: the thread's own, then the threads before it ----
		int lm = -1;
#pragma unroll
		for (int i = 0; i < SC_PER_T; i++)
			lm = coded[i] ? e0 + i : lm;
		const int incl_m = block_scan_incl<true>(lm, s_tmp);
		int prev = __shfl_up(incl_m, 1, 64);
		if (lane == 0)
			prev = wave ? s_tmp[16 + wave - 1] : -0x7fffffff;
		const int cmax = s_carry_max;
		prev = prev > cmax ? prev : cmax; // the last coded macroblock in front of e0
		const int tot_m = max(s_tmp[16 + nwaves - 1], cmax);
		// ---- skip runs and bits of the entries ----
		int r[SC_PER_T], bits[SC_PER_T];
		int lsum = 0;
		bool any_pcm = false;
#pragma unroll
		for (int i = 0; i < SC_PER_T; i++) {
			const int e = e0 + i;
			const bool trailer = e == n;
			r[i] = coded[i] || trailer ? e - 1 - prev : 0;
			bits[i] = coded[i] ? (is_p ? ue_len((uint32_t)r[i]) : 0) + (int)(mbb[i] & ~PCM_MB_BITS_FLAG)
				  : trailer ? ((is_p && r[i] > 0) ? ue_len((uint32_t)r[i]) : 0) + 1 // + rbsp stop bit
					    : 0;
			lsum += bits[i];
			any_pcm = any_pcm || (mbb[i] & PCM_MB_BITS_FLAG) != 0;
			prev = coded[i] ? e : prev;
		}
		// ---- positions ----
		int start[SC_PER_T], end[SC_PER_T];
		int tile_end;
		const int pos0 = s_carry_sum;
		if (!p.pcm) {
			const int incl = block_scan_incl<false>(lsum, s_tmp);
			int pos = pos0 + incl - lsum;
#pragma unroll
			for (int i = 0; i < SC_PER_T; i++) {
				start[i] = pos;
				pos += bits[i];
				end[i] = pos;
			}
			tile_end = pos0 + s_tmp[16 + nwaves - 1];
		} else {
			// I_PCM macroblocks: the samples start at the next byte boundary behind the header, so an entry is
			// no longer a length but a map of positions, pos -> pos + a, or pos -> align8(pos + a) + b behind
			// which everything is byte aligned again. Maps of that form compose to maps of that form (PosMap), and
			// composition is associative: an inclusive scan gives every entry the map of all entries up to it.
			PosMap loc[SC_PER_T + 1]; // loc[i]: the thread's entries before its i-th
			loc[0] = PosMap{ 0, 0, 0 };
#pragma unroll
			for (int i = 0; i < SC_PER_T; i++) {
				const bool pcm_e = (mbb[i] & PCM_MB_BITS_FLAG) != 0;
				loc[i + 1] = posmap_then(loc[i], PosMap{ bits[i], pcm_e ? 3072 : 0, pcm_e ? 1 : 0 });
			}
			const PosMap incl = block_scan_posmap(loc[SC_PER_T], s_pm);
			PosMap ex{ __shfl_up(incl.a, 1, 64), __shfl_up(incl.b, 1, 64), __shfl_up(incl.pcm, 1, 64) };
			if (lane == 0)
				ex = wave ? s_pm[16 + wave - 1] : PosMap{ 0, 0, 0 };
#pragma unroll
			for (int i = 0; i < SC_PER_T; i++) {
				start[i] = posmap_apply(posmap_then(ex, loc[i]), pos0);
				end[i] = posmap_apply(posmap_then(ex, loc[i + 1]), pos0);
			}
			tile_end = posmap_apply(s_pm[16 + nwaves - 1], pos0);
		}
		(void)any_pcm;
		// ---- out ----
#pragma unroll
		for (int i = 0; i < SC_PER_T; i++) {
			const int e = e0 + i;
			if (e > n)
				continue;
			const bool trailer = e == n;
			if (!p.ent_rec) { // (k_pack's and k_esc_*'s inputs)
				off[e] = (uint32_t)start[i];
				run[e] = (uint32_t)r[i];
			} else {
				// k_pack_out: everything it needs to know of an entry in one word, and for every chunk of PK_CHUNK output
				// bytes the entry that covers its first bit (an entry is at most 3088 bits: at most one chunk boundary)
				const uint32_t len = trailer ? 1u : (mbb[i] & ~PCM_MB_BITS_FLAG);
				// (run: 16 bits, a unit has at most 36864 macroblocks; len: 15 bits, 0 = a skipped macroblock -- a coded one has its mb_type)
				const uint32_t hi = (uint32_t)r[i] | (len & 0x7fffu) << 16 | ((mbb[i] & PCM_MB_BITS_FLAG) ? 1u : 0u) << 31;
				p.ent_rec[(size_t)unit * (p.unit_mb_cap + 2) + e] = (uint64_t)(uint32_t)start[i] | (uint64_t)hi << 32;
				if (end[i] > start[i]) {
					const int c = (start[i] + PK_BITS - 1) / PK_BITS;
					if (c * PK_BITS < end[i] && c < p.pk_chunks_cap)
						p.pk_first[(size_t)unit * p.pk_chunks_cap + c] = (uint32_t)e;
				}
			}
			if (trailer)
				p.unit_bits[unit] = (uint32_t)end[i];
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			s_carry_max = tot_m;
			s_carry_sum = tile_end;
		}
		__syncthreads();
	}
}

// ------------------------------------------------------------------------
// k_zero: clear the part of each unit's RBSP buffer that k_pack will OR into
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_zero(Params p)
{
	const int unit = blockIdx.y;
	const uint32_t nbytes = (p.unit_bits[unit] + 7) >> 3;
	const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * 16u;
	if (i < nbytes + 16 && i + 16 <= (uint32_t)p.rbsp_cap)
		*(u32x4 *)(p.rbsp + (size_t)unit * p.rbsp_cap + i) = u32x4{ 0, 0, 0, 0 };
}

// OR the low n bits of v into the RBSP at bit position pos (MSB first)
DEVI void or_bits(uint32_t *w, uint32_t cap_words, uint32_t pos, uint32_t v, int n)
{
	if (n <= 0)
		return;
	const uint32_t wi = pos >> 5, s = pos & 31;
	v = n >= 32 ? v : (v & ((1u << n) - 1u));
	if (wi + 1 >= cap_words)
		return; // overflow is reported by k_esc_scan through the size check
	if (s + n <= 32) {
		atomicOr(&w[wi], __builtin_bswap32(v << (32 - s - n)));
	} else {
		const int lo = s + n - 32; // bits spilling into the next word
		atomicOr(&w[wi], __builtin_bswap32(v >> lo));
		atomicOr(&w[wi + 1], __builtin_bswap32(v << (32 - lo)));
	}
}

// ------------------------------------------------------------------------
// k_pack: one thread per entry (header, macroblocks, trailer) of a unit
// ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack(Params p)
{
	const int unit = blockIdx.y;
	const int st = unit / p.slice_num, ls = unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, p.slice_first + ls, &r0, &r1);
	const int n = (r1 - r0) * p.mb_w;
	const int t = blockIdx.x * blockDim.x + threadIdx.x; // 0 = header, 1..n = MBs, n+1 = trailer
	if (t > n + 1)
		return;
	uint32_t *w = (uint32_t *)(p.rbsp + (size_t)unit * p.rbsp_cap);
	const uint32_t capw = (uint32_t)p.rbsp_cap >> 2;
	const uint32_t *off = p.bit_off + (size_t)unit * (p.unit_mb_cap + 2);
	const uint32_t *run = p.run_before + (size_t)unit * (p.unit_mb_cap + 2);
	const bool is_p = !p.is_idr;
	if (t == 0) {
		const uint64_t hb = p.hdr[unit].bits;
		const int hn = p.hdr[unit].nbits;
		if (hn > 32) {
			or_bits(w, capw, 0, (uint32_t)(hb >> 32), hn - 32);
			or_bits(w, capw, (uint32_t)(hn - 32), (uint32_t)hb, 32);
		} else {
			or_bits(w, capw, 0, (uint32_t)hb, hn);
		}
		return;
	}
	const int e = t - 1;
	uint32_t pos = off[e];
	if (e == n) { // trailer: pending skip run, then rbsp_trailing_bits
		if (is_p && run[e] > 0) {
			const uint32_t v = run[e] + 1;
			const int len = 2 * (32 - __clz(v)) - 1;
			or_bits(w, capw, pos, v, len);
			pos += len;
		}
		or_bits(w, capw, pos, 1, 1);
		return;
	}
	const size_t mbaddr = (size_t)st * p.n_mb + (size_t)r0 * p.mb_w + e;
	if (p.skip[mbaddr])
		return;
	if (is_p) {
		const uint32_t v = run[e] + 1;
		const int len = 2 * (32 - __clz(v)) - 1;
		or_bits(w, capw, pos, v, len);
		pos += len;
	}
	const uint32_t mbb = p.mb_bits[mbaddr], nbits = mbb & ~PCM_MB_BITS_FLAG;
	const uint32_t *slot = p.slot + mbaddr * SLOT_WORDS;
	const uint32_t full = nbits >> 5, rem = nbits & 31;
	for (uint32_t k = 0; k < full && k < (uint32_t)SLOT_WORDS; k++)
		or_bits(w, capw, pos + 32 * k, slot[k], 32);
	if (rem && full < (uint32_t)SLOT_WORDS)
		or_bits(w, capw, pos + 32 * full, slot[full] >> (32 - rem), (int)rem);
	if (mbb & PCM_MB_BITS_FLAG) {
		// I_PCM: pcm_alignment_zero_bits (the buffer is zero), then 256 luma, 64 Cb, 64 Cr samples from the source
		uint32_t bp = (pos + nbits + 7) & ~7u;
		const int my = r0 + e / p.mb_w, mx = e % p.mb_w;
		const uint8_t *sy = src_luma(p, st) + 16 * mx;
		const uint8_t *sc = src_chroma(p, st) + 16 * mx;
		for (int y = 0; y < 16; y++)
			for (int k = 0; k < 4; k++, bp += 32)
				or_bits(w, capw, bp, __builtin_bswap32(*(const uint32_t *)(sy + (size_t)min(16 * my + y, p.src_last_y) * p.cw + 4 * k)), 32);
		for (int c = 0; c < 2; c++)
			for (int y = 0; y < 8; y++) {
				const uint8_t *r = sc + (size_t)min(8 * my + y, p.src_last_c) * p.cw + c;
				for (int k = 0; k < 2; k++, bp += 32)
					or_bits(w, capw, bp, (uint32_t)r[8 * k] << 24 | (uint32_t)r[8 * k + 2] << 16 | (uint32_t)r[8 * k + 4] << 8 | r[8 * k + 6], 32);
			}
	}
}

// ---------------- emulation prevention ----------------
// zero-run state (0, 1, 2) of the escaping automaton just before byte `pos`
DEVI int zero_state_before(const uint8_t *rb, int pos)
{
	int run = 0;
	for (int i = pos - 1; i >= 0 && rb[i] == 0; i--)
		run++;
	return run == 0 ? 0 : ((run & 1) ? 1 : 2);
}

// The escaping automaton over one ESC_CHUNK of RBSP held in registers. EMIT = write the
// escaped bytes to o (byte stores: the output is shifted by a running, unaligned amount).
template <bool EMIT>
DEVI uint32_t esc_chunk(const uint8_t *rb, int start, int end, uint8_t *o)
{
	static_assert(ESC_CHUNK == 32, "two 16-byte loads per chunk");
	const u32x4 a = *(const u32x4 *)(rb + start), c = *(const u32x4 *)(rb + start + 16);
	const uint32_t w[8] = { a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w };
	int z = zero_state_before(rb, start);
	uint32_t cnt = 0;
#pragma unroll
	for (int i = 0; i < ESC_CHUNK; i++) {
		if (start + i < end) {
			const uint32_t v = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
			if (z == 2 && v <= 3) {
				cnt++;
				z = 0;
				if (EMIT)
					*o++ = 3;
			}
			if (EMIT)
				*o++ = (uint8_t)v;
			z = v == 0 ? z + 1 : 0;
		}
	}
	return cnt;
}

__global__ __launch_bounds__(256) void k_esc_count(Params p)
{
	const int unit = blockIdx.y;
	const int nbytes = (int)((p.unit_bits[unit] + 7) >> 3);
	const int chunk = blockIdx.x * blockDim.x + threadIdx.x;
	const int start = chunk * ESC_CHUNK;
	if (chunk >= p.esc_chunks_cap || start >= nbytes + ESC_CHUNK) // one zero entry past the end is all the scan reads
		return;
	uint32_t cnt = 0;
	if (start < nbytes) {
		const uint8_t *rb = p.rbsp + (size_t)unit * p.rbsp_cap;
		cnt = esc_chunk<false>(rb, start, start + ESC_CHUNK < nbytes ? start + ESC_CHUNK : nbytes, nullptr);
	}
	p.esc_cnt[(size_t)unit * p.esc_chunks_cap + chunk] = cnt;
}

// One block per unit: exclusive prefix of its escape counts. The block that finishes last
// lays the NAL units out: sizes, compact output offsets (stream-major, each stream preceded
// by its SPS/PPS prefix), total and overflow flag.
__global__ __launch_bounds__(1024) void k_esc_scan(Params p)
{
	__shared__ int s_tmp[32];
	__shared__ int s_carry;
	__shared__ int s_last;
	const int n_units = p.n_streams * p.slice_num;
	const int unit = blockIdx.x;
	{
		const int nbytes = (int)((p.unit_bits[unit] + 7) >> 3);
		const int chunks = (nbytes + ESC_CHUNK - 1) / ESC_CHUNK;
		uint32_t *cnt = p.esc_cnt + (size_t)unit * p.esc_chunks_cap;
		if (threadIdx.x == 0)
			s_carry = 0;
		__syncthreads();
		for (int base = 0; base < chunks; base += SCAN_THREADS) {
			const int c = base + threadIdx.x;
			const int v = c < chunks ? (int)cnt[c] : 0;
			const int incl = block_scan_incl<false>(v, s_tmp);
			const int carry = s_carry;
			if (c < chunks)
				cnt[c] = (uint32_t)(carry + incl - v);
			__syncthreads();
			if (threadIdx.x == SCAN_THREADS - 1)
				s_carry = carry + incl;
			__syncthreads();
		}
		if (threadIdx.x == 0) {
			p.nal_size[unit] = 5u + (uint32_t)nbytes + (uint32_t)s_carry;
			__threadfence();
			s_last = atomicAdd(&p.stream_size[p.n_streams + 2], 1u) == (uint32_t)n_units - 1;
		}
		__syncthreads();
	}
	if (!s_last || threadIdx.x != 0)
		return;
	__threadfence();
	uint32_t pos = 0, flags = 0;
	for (int u = 0; u < n_units; u++) {
		const int st = u / p.slice_num, ls = u % p.slice_num;
		const uint32_t size = __hip_atomic_load(&p.nal_size[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		const uint32_t nbytes = (__hip_atomic_load(&p.unit_bits[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 7) >> 3;
		uint32_t start_of_stream = 0;
		if (ls == 0) {
			start_of_stream = pos;
			p.stream_size[st] = pos; // start of the stream, fixed up below
			pos += p.prefix_len[st];
		}
		(void)start_of_stream;
		p.nal_off[u] = pos;
		pos += size;
		if (ls == p.slice_num - 1)
			p.stream_size[st] = pos - p.stream_size[st];
		if (pos > (uint32_t)p.out_cap || nbytes + 8u > (uint32_t)p.rbsp_cap)
			flags |= 2u;
	}
	p.stream_size[p.n_streams] = pos;
	if (flags)
		atomicOr(&p.stream_size[p.n_streams + 1], flags);
}

__global__ __launch_bounds__(256) void k_esc_write(Params p)
{
	const int unit = blockIdx.y;
	const int nbytes = (int)((p.unit_bits[unit] + 7) >> 3);
	const int chunk = blockIdx.x * blockDim.x + threadIdx.x;
	const int start = chunk * ESC_CHUNK;
	if (start >= nbytes || (p.stream_size[p.n_streams + 1] & 2u))
		return;
	const uint8_t *rb = p.rbsp + (size_t)unit * p.rbsp_cap;
	uint8_t *o = p.out + p.nal_off[unit];
	if (chunk == 0) { // 4-byte start code on every NAL (cedar.c:875-878) + header byte
		o[0] = 0;
		o[1] = 0;
		o[2] = 0;
		o[3] = 1;
		o[4] = (uint8_t)((p.nal_ref_idc << 5) | p.nal_type);
	}
	o += 5 + start + p.esc_cnt[(size_t)unit * p.esc_chunks_cap + chunk];
	esc_chunk<true>(rb, start, start + ESC_CHUNK < nbytes ? start + ESC_CHUNK : nbytes, o);
}


// ------------------------------------------------------------------------
// k_pack_out (round 4): everything behind k_scan in ONE launch -- bit-granular concatenation, emulation prevention, start
// code + NAL header, compaction of all NAL units into the caller's pinned buffer, sizes and the completion flag. It takes
// the place of k_zero, k_pack, k_esc_count, k_esc_scan, k_esc_write and k_copy_out, and of the RBSP buffer they met in
// (4.4 MB of atomic ORs into a zeroed buffer to place 0.2 MB of bits; CEDAR_ENTROPY_FUSED=0 keeps that path).
//
// A workgroup owns PK_CHUNK bytes of one unit's RBSP. It builds them in LDS: k_scan left, per chunk, the entry that covers
// its first bit (pk_first) and per entry one word with its bit position, skip run and length (ent_rec); the threads take the
// chunk's entries 256 at a time and OR their slot words -- clipped to the chunk -- into LDS. Escaping is the automaton
// "two zero bytes, then a byte <= 3 gets a 03 in front", whose state before a byte only depends on the zero run in front of
// it: every thread runs its 8 bytes from each of the three states, a (count, state-after) table per thread, and tables
// compose associatively -- a scan inside the workgroup, and the workgroup's total goes out as ONE tagged 64-bit word
// (pk_agg). A chunk learns the escapes in front of it and its state by composing the words of the chunks before it in its
// unit, a unit's last chunk publishes the unit's total (pk_unit), and a NAL's place in the output is the sum of what is in
// front of it: a workgroup only ever waits for workgroups with a lower index, which the hardware started no later (each XCD
// takes its share of a launch in order), and the words carry the frame's tag, so nothing needs clearing. The escaped bytes
// are laid out in LDS congruent to their address modulo 16 and leave as 16-byte stores (byte stores at a chunk's ragged
// ends). The workgroup that counts out last writes the sizes and raises the frame's flag.
// ------------------------------------------------------------------------
constexpr int PK_THREADS = 256;
constexpr int PK_PER_T = PK_CHUNK / PK_THREADS; // 8 bytes of RBSP per thread
static_assert(PK_PER_T == 8, "a thread's bytes are two LDS words");

// table of the escaping automaton over a run of bytes: counts c[s] (16 bits each) and states after o[s] (2 bits each) for
// the three states s before it
constexpr uint64_t PK_TAB_ID = (uint64_t)0 << 48 | (uint64_t)1 << 50 | (uint64_t)2 << 52;
DEVI uint32_t tab_c(uint64_t t, int s) { return (uint32_t)(t >> (16 * s)) & 0xffffu; }
DEVI int tab_o(uint64_t t, int s) { return (int)(t >> (48 + 2 * s)) & 3; }
DEVI uint64_t tab_then(uint64_t a, uint64_t b) // a, then b
{
	uint64_t r = 0;
#pragma unroll
	for (int s = 0; s < 3; s++) {
		const int oa = tab_o(a, s);
		r |= (uint64_t)((tab_c(a, s) + tab_c(b, oa)) & 0xffffu) << (16 * s) | (uint64_t)tab_o(b, oa) << (48 + 2 * s);
	}
	return r;
}
DEVI uint64_t shfl_up64(uint64_t v, int off)
{
	const uint32_t lo = (uint32_t)__shfl_up((int)(uint32_t)v, off, 64), hi = (uint32_t)__shfl_up((int)(uint32_t)(v >> 32), off, 64);
	return (uint64_t)hi << 32 | lo;
}
DEVI uint64_t shfl64(uint64_t v, int lane)
{
	const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, lane, 64), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), lane, 64);
	return (uint64_t)hi << 32 | lo;
}
// inclusive scan of tables over the lanes of a wave, lane order = byte order
DEVI uint64_t tab_wave_scan(uint64_t v, int lane)
{
#pragma unroll
	for (int off = 1; off < 64; off <<= 1) {
		const uint64_t o = shfl_up64(v, off);
		if (lane >= off)
			v = tab_then(o, v);
	}
	return v;
}
// a chunk's table as it travels: count from state 0 in 12 bits, the other two as differences (-2..5), the states after
DEVI uint32_t tab_to_word(uint64_t t)
{
	const int c0 = (int)tab_c(t, 0), d1 = (int)tab_c(t, 1) - c0 + 2, d2 = (int)tab_c(t, 2) - c0 + 2;
	return (uint32_t)c0 | (uint32_t)(d1 & 7) << 12 | (uint32_t)(d2 & 7) << 15 | (uint32_t)tab_o(t, 0) << 18 | (uint32_t)tab_o(t, 1) << 20 | (uint32_t)tab_o(t, 2) << 22;
}
DEVI uint64_t tab_from_word(uint32_t w)
{
	const uint32_t c0 = w & 0xfffu, c1 = c0 + ((w >> 12) & 7u) - 2u, c2 = c0 + ((w >> 15) & 7u) - 2u;
	return (uint64_t)c0 | (uint64_t)(c1 & 0xffffu) << 16 | (uint64_t)(c2 & 0xffffu) << 32 | (uint64_t)((w >> 18) & 3u) << 48 | (uint64_t)((w >> 20) & 3u) << 50 |
	       (uint64_t)((w >> 22) & 3u) << 52;
}

// wait for a tagged word (upper half = this frame's tag); bounded like every hand-off of the path
DEVI uint32_t pk_wait_word(const uint64_t *w, uint32_t tag, const Params &p, bool &dead)
{
	uint64_t v = xload64(w);
	uint32_t spins = 0;
	uint64_t t0 = 0;
	while ((uint32_t)(v >> 32) != tag && !dead) {
		if (wait_expired(++spins, t0)) { // (its own code: the word it waited for never came)
			__hip_atomic_store(p.wf_err, 64u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			__hip_atomic_store(p.abort_all, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
			dead = true;
			break;
		}
		if ((spins & 63u) == 0 && __hip_atomic_load(p.abort_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { // another wait of the session ran out
			dead = true;
			break;
		}
		__builtin_amdgcn_s_sleep(2);
		v = xload64(w);
	}
	return (uint32_t)v;
}

// OR the low n bits of v (n <= 32) into the chunk image at bit position rel (relative to the chunk; may lie outside)
DEVI void pk_or(uint32_t *img, int rel, uint32_t v, int n)
{
	if (n <= 0 || rel >= PK_BITS || rel + n <= 0)
		return;
	if (n < 32)
		v &= (1u << n) - 1u;
	if (rel < 0) { // the front is another chunk's
		n += rel;
		v &= (1u << n) - 1u; // (n < 32 here)
		rel = 0;
	}
	if (rel + n > PK_BITS) { // so is the tail
		const int cut = rel + n - PK_BITS;
		v >>= cut;
		n -= cut;
	}
	const int wi = rel >> 5, sft = rel & 31;
	if (sft + n <= 32) {
		atomicOr(&img[wi], __builtin_bswap32(v << (32 - sft - n)));
	} else {
		const int lo = sft + n - 32;
		atomicOr(&img[wi], __builtin_bswap32(v >> lo));
		atomicOr(&img[wi + 1], __builtin_bswap32(v << (32 - lo))); // (img has a word of slack)
	}
}

__global__ __launch_bounds__(PK_THREADS) void k_pack_out(Params p, uint8_t *host_out, uint32_t *host_sizes)
{
	__shared__ __attribute__((aligned(16))) uint32_t s_img[PK_CHUNK / 4 + 4];
	__shared__ __attribute__((aligned(16))) uint8_t s_o[16 + 5 + PK_CHUNK + PK_CHUNK / 2 + 32];
	__shared__ uint64_t s_wt[PK_THREADS / 64];
	__shared__ uint32_t s_pre[6]; // escapes in front of the chunk, state before it, NAL offset, chunks of the launch, dead, last
	__shared__ int s_job[4]; // the chunk this workgroup does next: unit, chunk, chunks of the launch; -1: none left
	const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
	const int n_units = p.n_streams * p.slice_num;
	const uint32_t tag = p.db_tag;
	const bool is_p = !p.is_idr;
	bool dead = false;
	// Chunks are taken by TICKET, in the order the workgroups get to run: ticket k is chunk k of the launch in unit-major
	// order, and a chunk only waits for chunks with lower tickets -- which were taken by workgroups that are RUNNING, so the
	// wait ends whatever else fills the chip. (The first version gave workgroup (unit, chunk) its chunk by index and relied
	// on every XCD starting its share of a launch in order: true, but two such launches side by side -- two encoder processes
	// on one GPU, 2 x 32 streams -- can each fill an XCD with workgroups that wait for one the OTHER launch leaves no room
	// for: -EIO after 3 s in tests/test_bench_contract.py's two-rank rehearsal.) A workgroup that has done its chunk takes the
	// next ticket: the launch needs no workgroup per possible chunk, only enough of them to keep the chip busy.
	for (;;) {
	if (wave == 0) {
		int ticket = 0;
		if (lane == 0)
			ticket = (int)__hip_atomic_fetch_add(&p.stream_size[p.n_streams + 3], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		ticket = __builtin_amdgcn_readfirstlane(ticket);
		int before = 0, found_u = -1, found_c = 0; // chunks of the units in front; the ticket's unit and chunk
		for (int base = 0; base < n_units; base += 64) {
			const int u2 = base + lane;
			const uint32_t nb2 = u2 < n_units ? (p.unit_bits[u2] + 7) >> 3 : 0u;
			const int nc2 = (int)min((nb2 + PK_CHUNK - 1) / PK_CHUNK, (uint32_t)p.pk_chunks_cap);
			int incl = nc2;
#pragma unroll
			for (int off = 1; off < 64; off <<= 1) {
				const int o = __shfl_up(incl, off, 64);
				if (lane >= off)
					incl += o;
			}
			const int lo = before + incl - nc2; // first ticket of unit u2
			const uint64_t hit = __ballot(ticket >= lo && ticket < lo + nc2);
			if (hit && found_u < 0) {
				const int l = __builtin_ctzll(hit);
				found_u = base + l;
				found_c = ticket - __shfl(lo, l, 64);
			}
			before += __shfl(incl, 63, 64);
		}
		if (lane == 0) {
			s_job[0] = found_u;
			s_job[1] = found_c;
			s_job[2] = before; // chunks of the launch
		}
	}
	__syncthreads();
	const int unit = s_job[0], c = s_job[1];
	const uint32_t all_chunks = (uint32_t)s_job[2];
	if (unit < 0)
		return; // (no chunk left: uniform)
	const uint32_t nbytes = (p.unit_bits[unit] + 7) >> 3;
	const int n_chunks = (int)min((nbytes + PK_CHUNK - 1) / PK_CHUNK, (uint32_t)p.pk_chunks_cap);
	const int st = unit / p.slice_num, ls = unit % p.slice_num;
	int r0, r1;
	slice_rows(p.mb_h, p.slice_count, p.slice_first + ls, &r0, &r1);
	const int n = (r1 - r0) * p.mb_w;
	const int bit0 = c * PK_BITS;                       // first bit of the chunk
	const int len_c = (int)min((uint32_t)PK_CHUNK, nbytes - (uint32_t)c * PK_CHUNK); // its bytes

	// ---- the chunk's RBSP bytes, in LDS ----
	for (int i = t; i < PK_CHUNK / 4 + 4; i += PK_THREADS)
		s_img[i] = 0;
	__syncthreads();
	if (c == 0 && t == 0) { // the slice header
		const uint64_t hb = p.hdr[unit].bits;
		const int hn = p.hdr[unit].nbits;
		if (hn > 32) {
			pk_or(s_img, 0, (uint32_t)(hb >> 32), hn - 32);
			pk_or(s_img, hn - 32, (uint32_t)hb, 32);
		} else {
			pk_or(s_img, 0, (uint32_t)hb, hn);
		}
	}
	{
		const uint64_t *rec = p.ent_rec + (size_t)unit * (p.unit_mb_cap + 2);
		const int e_lo = c == 0 ? 0 : (int)p.pk_first[(size_t)unit * p.pk_chunks_cap + c];
		const size_t mb_base = (size_t)st * p.n_mb + (size_t)r0 * p.mb_w;
		constexpr int PK_AHEAD = 4; // entries a thread has in flight: their records, then the first two words of their slots
		bool done = false;
		for (int eb = e_lo + t; eb <= n && !done; eb += PK_AHEAD * PK_THREADS) {
			uint64_t rcs[PK_AHEAD];
			uint32_t s0[PK_AHEAD], s1[PK_AHEAD];
#pragma unroll
			for (int j = 0; j < PK_AHEAD; j++) {
				const int e = eb + j * PK_THREADS;
				rcs[j] = e <= n ? rec[e] : ~0ull; // (a position behind every chunk)
			}
#pragma unroll
			for (int j = 0; j < PK_AHEAD; j++) {
				const int e = eb + j * PK_THREADS;
				const uint32_t hi = (uint32_t)(rcs[j] >> 32), len = (hi >> 16) & 0x7fffu;
				const bool want = e < n && len != 0 && (int)(uint32_t)rcs[j] - bit0 < PK_BITS;
				const uint32_t *slot = p.slot + (mb_base + (want ? e : 0)) * SLOT_WORDS;
				s0[j] = want ? slot[0] : 0u;
				s1[j] = want && len > 32 ? slot[1] : 0u;
			}
#pragma unroll
			for (int j = 0; j < PK_AHEAD; j++) {
				const int e = eb + j * PK_THREADS;
				const uint64_t rc = rcs[j];
				int pos = (int)(uint32_t)rc - bit0;
				if (e > n || pos >= PK_BITS) { // (positions ascend: nothing further on is this chunk's)
					done = true;
					break;
				}
				const uint32_t hi = (uint32_t)(rc >> 32), run = hi & 0xffffu, len = (hi >> 16) & 0x7fffu;
				if (len == 0)
					continue; // a skipped macroblock
				if (is_p && (e < n || run > 0)) { // mb_skip_run in front of a coded macroblock, or pending at the end of the slice
					const uint32_t v = run + 1;
					const int l = 2 * (32 - __clz(v)) - 1;
					pk_or(s_img, pos, v, l);
					pos += l;
				}
				if (e == n) { // rbsp_trailing_bits: the stop bit (the image is zero behind it)
					pk_or(s_img, pos, 1, 1);
					continue;
				}
				const uint32_t *slot = p.slot + (mb_base + e) * SLOT_WORDS;
				const int full = (int)(len >> 5), rem = (int)(len & 31);
				if (pos + (int)len > 0) { // (an entry that ends in front of the chunk has nothing for it; pk_first never points at one)
					int k = pos < 0 ? (-pos) >> 5 : 0; // words wholly in front of the chunk
					for (; k < full && k < SLOT_WORDS && pos + 32 * k < PK_BITS; k++)
						pk_or(s_img, pos + 32 * k, k == 0 ? s0[j] : k == 1 ? s1[j] : slot[k], 32);
					if (rem && full < SLOT_WORDS)
						pk_or(s_img, pos + 32 * full, (full == 0 ? s0[j] : full == 1 ? s1[j] : slot[full]) >> (32 - rem), rem);
				}
				if (hi >> 31) {
					// I_PCM: pcm_alignment_zero_bits (the image is zero), then 256 luma, 64 Cb, 64 Cr samples from the source
					int bp = (pos + (int)len + 7) & ~7; // (bit0 is a multiple of 8)
					const int my = r0 + e / p.mb_w, mx = e % p.mb_w;
					const uint8_t *sy = src_luma(p, st) + 16 * mx;
					const uint8_t *sc = src_chroma(p, st) + 16 * mx;
					for (int y = 0; y < 16; y++)
						for (int k = 0; k < 4; k++, bp += 32)
							pk_or(s_img, bp, __builtin_bswap32(*(const uint32_t *)(sy + (size_t)min(16 * my + y, p.src_last_y) * p.cw + 4 * k)), 32);
					for (int cc = 0; cc < 2; cc++)
						for (int y = 0; y < 8; y++) {
							const uint8_t *r = sc + (size_t)min(8 * my + y, p.src_last_c) * p.cw + cc;
							for (int k = 0; k < 2; k++, bp += 32)
								pk_or(s_img, bp, (uint32_t)r[8 * k] << 24 | (uint32_t)r[8 * k + 2] << 16 | (uint32_t)r[8 * k + 4] << 8 | r[8 * k + 6], 32);
						}
				}
			}
		}
	}
	__syncthreads();

	// ---- the escaping automaton over this thread's 8 bytes, from each of the three states ----
	const uint32_t w0 = s_img[2 * t], w1 = s_img[2 * t + 1];
	const int nb = min(max(len_c - PK_PER_T * t, 0), PK_PER_T);
	uint64_t tab = 0;
#pragma unroll
	for (int s = 0; s < 3; s++) {
		int z = s;
		uint32_t cnt = 0;
#pragma unroll
		for (int i = 0; i < PK_PER_T; i++)
			if (i < nb) {
				const uint32_t v = ((i < 4 ? w0 : w1) >> (8 * (i & 3))) & 0xffu;
				if (z == 2 && v <= 3) {
					cnt++;
					z = 0;
				}
				z = v == 0 ? z + 1 : 0;
			}
		tab |= (uint64_t)cnt << (16 * s) | (uint64_t)z << (48 + 2 * s);
	}
	const uint64_t incl = tab_wave_scan(tab, lane);
	if (lane == 63)
		s_wt[wave] = incl;
	__syncthreads();
	uint64_t wpre = PK_TAB_ID, total = PK_TAB_ID;
#pragma unroll
	for (int w = 0; w < PK_THREADS / 64; w++) {
		if (w < wave)
			wpre = tab_then(wpre, s_wt[w]);
		total = tab_then(total, s_wt[w]);
	}
	uint64_t excl = shfl_up64(incl, 1);
	if (lane == 0)
		excl = PK_TAB_ID;
	excl = tab_then(wpre, excl); // the bytes of the chunk in front of this thread's
	uint64_t *agg = p.pk_agg + (size_t)unit * p.pk_chunks_cap;
	if (t == 0)
		xstore64(&agg[c], (uint64_t)tag << 32 | tab_to_word(total));

	// ---- what lies in front: the chunks before this one in its unit, the units before this one ----
	if (wave == 0) {
		uint64_t pre = PK_TAB_ID;
		for (int base = 0; base < c; base += 64) {
			const int i = base + lane;
			uint64_t v = PK_TAB_ID;
			if (i < c)
				v = tab_from_word(pk_wait_word(&agg[i], tag, p, dead));
			v = tab_wave_scan(v, lane);
			pre = tab_then(pre, shfl64(v, 63));
		}
		const uint32_t esc_before = tab_c(pre, 0); // (a NAL's payload starts in state 0: its header byte is not zero)
		const int s_in = tab_o(pre, 0);
		const uint32_t esc_unit = esc_before + tab_c(total, s_in);
		if (c == n_chunks - 1 && lane == 0)
			xstore64(&p.pk_unit[unit], (uint64_t)tag << 32 | esc_unit);
		uint32_t front = 0, chunks = 0;
		for (int base = 0; base < n_units; base += 64) {
			const int u2 = base + lane;
			if (u2 < n_units) {
				const uint32_t nb2 = (p.unit_bits[u2] + 7) >> 3;
				chunks += min((nb2 + PK_CHUNK - 1) / PK_CHUNK, (uint32_t)p.pk_chunks_cap);
				if (u2 < unit)
					front += 5u + nb2 + pk_wait_word(&p.pk_unit[u2], tag, p, dead) + (u2 % p.slice_num == 0 ? p.prefix_len[u2 / p.slice_num] : 0u);
			}
		}
#pragma unroll
		for (int off = 32; off > 0; off >>= 1) {
			front += (uint32_t)__shfl_xor((int)front, off, 64);
			chunks += (uint32_t)__shfl_xor((int)chunks, off, 64);
		}
		const bool any_dead = __any(dead);
		if (lane == 0) {
			s_pre[0] = esc_before;
			s_pre[1] = (uint32_t)s_in;
			s_pre[2] = front + (ls == 0 ? p.prefix_len[st] : 0u);
			s_pre[3] = chunks;
			s_pre[4] = any_dead ? 1u : 0u;
		}
	}
	__syncthreads();
	const uint32_t esc_before = s_pre[0];
	const int s_in = (int)s_pre[1];
	const uint32_t nal_off = s_pre[2];
	const bool failed = s_pre[4] != 0;

	// ---- escaped bytes into LDS, congruent to their address modulo 16; then out ----
	const uint32_t out0 = nal_off + (c == 0 ? 0u : 5u + (uint32_t)c * PK_CHUNK + esc_before); // first byte this chunk writes
	const uint32_t head = c == 0 ? 5u : 0u;
	const uint32_t out_len = head + (uint32_t)len_c + tab_c(total, s_in);
	const uint32_t a = (uint32_t)((uintptr_t)(host_out + out0) & 15u);
	if (c == 0 && t < 5) // 4-byte start code on every NAL (cedar.c:875-878) + header byte
		s_o[a + t] = t == 3 ? 1 : t == 4 ? (uint8_t)((p.nal_ref_idc << 5) | p.nal_type) : 0;
	{
		int z = tab_o(excl, s_in);
		uint8_t *o = s_o + a + head + PK_PER_T * t + tab_c(excl, s_in);
#pragma unroll
		for (int i = 0; i < PK_PER_T; i++)
			if (i < nb) {
				const uint32_t v = ((i < 4 ? w0 : w1) >> (8 * (i & 3))) & 0xffu;
				if (z == 2 && v <= 3) {
					*o++ = 3;
					z = 0;
				}
				*o++ = (uint8_t)v;
				z = v == 0 ? z + 1 : 0;
			}
	}
	__syncthreads();
	const bool fits = out0 + out_len <= (uint32_t)p.out_cap && !failed;
	if (fits) {
		uint8_t *g = host_out + out0 - a; // 16-byte aligned
		for (uint32_t j = t; 16 * j < a + out_len; j += PK_THREADS) {
			const uint32_t lo = 16 * j, hi16 = lo + 16;
			// (how the bytes get there: see the count-out below)
			if (lo >= a && hi16 <= a + out_len) {
				const u32x4 v = *(const u32x4 *)(s_o + lo);
				if (p.pk_deliver == 0)
					*(u32x4 *)(g + lo) = v;
				else // 16 bytes written through to the host (system scope): the line does not stay in this XCD's L2
					asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(g + lo), "v"(v) : "memory");
			} else {
				for (uint32_t b = max(lo, a); b < min(hi16, a + out_len); b++) {
					if (p.pk_deliver == 0)
						g[b] = s_o[b];
					else
						__hip_atomic_store(g + b, s_o[b], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
				}
			}
		}
	} else if (t == 0) {
		atomicOr(&p.stream_size[p.n_streams + 1], 2u);
	}

	// ---- count out; the last workgroup of the launch lays the sizes down and raises the frame's flag ----
	// WHEN ARE THE BYTES ON THE HOST? The GPU caches pinned host memory like any other: a plain store sits in the L2 of the
	// XCD that made it until that L2 is written back. Three ways, all measured (32 x 1080p / 8 x 1080p / one stream, M MB/s):
	//  * a release fence in every workgroup before it counts out (what k_copy_out does): it writes the XCD's WHOLE L2 back
	//    -- beside a batch's k_inter and deblocking that is megabytes of other kernels' dirty lines per workgroup: 144 / 129 /
	//    51. Without it, or with "coherent" host memory and s_waitcnt vmcnt(0) alone, or with a read-back of the chunk's last
	//    byte, the flag overtook the bytes: a pipelined session delivered whole chunks of the frame 16 before
	//    (tools/dev/pk_diff.py), at 194 / 140 / 57 -- the price list's "plain payload + vmcnt(0), no release: stale";
	//  * pk_deliver 1: WRITE-THROUGH stores (global_store_dwordx4 sc0 sc1: 16 bytes a lane, a wave writes whole lines; 8-byte
	//    ones crawl at 27 MB/s over the link) and s_waitcnt vmcnt(0): delivered when acknowledged, the last workgroup raises
	//    the flag itself. 168 / 124 / 55: every workgroup sits out a round trip to the host on the entropy stream;
	//  * pk_deliver 0: plain stores, nobody waits, and the FLAG comes from the next kernel of the stream (k_flag): a kernel's
	//    end is a release for everything it stored. 195 / 139 / 57. What sessions whose frames run back to back always did.
	if (p.pk_deliver == 1)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	uint32_t *cnt = p.stream_size + p.n_streams + 2;
	if (t == 0 && __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == all_chunks - 1) {
	uint32_t pos = 0, stream_start = 0, flags = __hip_atomic_load(&p.stream_size[p.n_streams + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	for (int u = 0; u < n_units; u++) { // stream-major: every stream's NAL units back to back behind its SPS / PPS / AUD prefix
		const int st2 = u / p.slice_num, ls2 = u % p.slice_num;
		const uint32_t nb2 = (p.unit_bits[u] + 7) >> 3;
		if (ls2 == 0) {
			stream_start = pos;
			pos += p.prefix_len[st2];
		}
		const uint64_t uw = xload64(&p.pk_unit[u]);
		const uint32_t size = 5u + nb2 + ((uint32_t)(uw >> 32) == tag ? (uint32_t)uw : 0u);
		p.nal_off[u] = pos;
		p.nal_size[u] = size;
		pos += size;
		if (ls2 == p.slice_num - 1)
			__hip_atomic_store(&host_sizes[st2], pos - stream_start, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		if (pos > (uint32_t)p.out_cap || (nb2 + PK_CHUNK - 1) / PK_CHUNK > (uint32_t)p.pk_chunks_cap || (uint32_t)(uw >> 32) != tag)
			flags |= 2u;
	}
	__hip_atomic_store(&host_sizes[p.n_streams], pos, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	__hip_atomic_store(&host_sizes[p.n_streams + 1], flags, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	if (p.cut_votes) { // scene cuts (one stream, one cohort): the picture's votes (the analysis is over: the entropy stream waited for it)
		__hip_atomic_store(&host_sizes[p.n_streams + 2], p.cut_votes[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
		p.cut_votes[0] = 0; // ... and the slot's next P picture starts from zero
	}
	for (int i = 0; i < p.n_streams + 3; i++) // the size words are left zeroed for the lane's next frame (the ticket word: k_scan)
		__hip_atomic_store(&p.stream_size[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the sizes are on the host before the flag is)
	if (p.done_flag && p.pk_deliver == 1)
		__hip_atomic_store(p.done_flag, p.done_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
	__syncthreads(); // (the LDS images and the job words are the next chunk's now)
	}
}

int Launch::entropy(const Params &p, void *stream)
{
	hipStream_t s = (hipStream_t)stream;
	const int n_units = p.n_streams * p.slice_num;
	const int mbs = p.row_num * p.mb_w;
	// Few macroblocks: k_cavlc2 (a lane per block, short chains; 1080p x 1: 0.04 ms against
	// 0.25 ms). Many: k_cavlc (a thread per macroblock does ~1/3 of the instructions per
	// macroblock; 1080p x 32: 0.40 ms against 1.38 ms).
	static int version = -1;
	if (version < 0) {
		const char *e = getenv("CEDAR_CAVLC_V");
		version = e ? atoi(e) : 0;
	}
	if (version == 2 || (version == 0 && (long)mbs * p.n_streams <= 40000))
		hipLaunchKernelGGL(k_cavlc2, dim3((mbs + 1) / 2, p.n_streams), dim3(64), 0, s, p);
	else {
		// (a wave of fewer macroblocks diverges less and leaves room for more waves: CEDAR_CAVLC_LANES, development)
		static int lanes = 0;
		if (!lanes) {
			const char *e = getenv("CEDAR_CAVLC_LANES");
			lanes = e ? atoi(e) : 64;
			if (lanes != 8 && lanes != 16 && lanes != 32)
				lanes = 64;
		}
		hipLaunchKernelGGL(k_cavlc, dim3((mbs + lanes - 1) / lanes, p.n_streams), dim3(lanes), 0, s, p);
	}
	hipLaunchKernelGGL(k_scan, dim3(n_units), dim3(n_units > 8 ? 256 : 1024), 0, s, p);
	if (p.ent_rec) // the fused back end: Launch::entropy_out adds k_pack_out
		return (int)hipGetLastError();
	hipLaunchKernelGGL(k_zero, dim3((p.rbsp_cap / 16 + 255) / 256, n_units), dim3(256), 0, s, p);
	hipLaunchKernelGGL(k_pack, dim3((p.unit_mb_cap + 2 + 255) / 256, n_units), dim3(256), 0, s, p);
	hipLaunchKernelGGL(k_esc_count, dim3((p.esc_chunks_cap + 255) / 256, n_units), dim3(256), 0, s, p);
	hipLaunchKernelGGL(k_esc_scan, dim3(n_units), dim3(n_units > 8 ? 256 : 1024), 0, s, p);
	hipLaunchKernelGGL(k_esc_write, dim3((p.esc_chunks_cap + 255) / 256, n_units), dim3(256), 0, s, p);
	return (int)hipGetLastError();
}

// cavlc, scan and -- the fused back end -- k_pack_out; or the round-3 chain with k_copy_out behind it
int Launch::entropy_out(const Params &p, void *stream, uint8_t *host_out, uint32_t *host_sizes)
{
	if (entropy(p, stream))
		return -1;
	if (!p.ent_rec)
		return copy_out(p, stream, host_out, host_sizes);
	// as many workgroups as the launch is expected to have chunks (pk_launch_chunks per unit), at most four to a CU: they take
	// chunks by ticket until none is left, so neither too few nor too many is wrong
	const int n_units = p.n_streams * p.slice_num;
	const long want = (long)n_units * (p.pk_launch_chunks > 0 ? p.pk_launch_chunks : 1);
	const unsigned grid = (unsigned)(want < 1 ? 1 : want > 1024 ? 1024 : want);
	hipLaunchKernelGGL(k_pack_out, dim3(grid), dim3(PK_THREADS), 0, (hipStream_t)stream, p, host_out, host_sizes);
	if (p.done_flag && p.pk_deliver == 0) // the kernel's end delivers its bytes: the flag is the next kernel's to raise
		return flag(stream, p.done_flag, p.done_tag);
	return (int)hipGetLastError();
}

} // namespace cedar
