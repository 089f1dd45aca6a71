// cedarx_h264_encoder_amd/csrc/enc_cavlc_len.h
// The LENGTH of residual_block_cavlc() (H.264 7.3.5.3.2 / 9.2) for one block, for the I_PCM decision of the
// analysis kernels (enc_k_parallel.hip): a macroblock whose CAVLC bits could pass the 3200 of A.3.1 is coded
// I_PCM. Level codes with their suffixLength, total_zeros and run_before are counted exactly as
// enc_k_entropy.hip writes them; coeff_token -- whose table depends on the neighbours' counts, which an
// analysis wave does not have -- at the longest of the four tables. One lane per block; the rare path of its
// kernels (a macroblock with fewer than 64 levels cannot get there).
#pragma once

#include "enc_device.h"
#include "enc_tables.h"

namespace cedar {

// lv: the block's levels in scan order (int16, LDS or registers' spill: any address space), entries [0, maxc)
DEVI int cavlc_block_len(const int16_t *lv, int maxc, bool chroma_dc)
{
	uint32_t mask = 0;
	for (int i = 0; i < maxc; i++)
		mask |= lv[i] != 0 ? 1u << i : 0u;
	const int n = __popc(mask);
	int t1 = 0;
	{
		uint32_t m = mask;
		while (m && t1 < 3) {
			const int hi = 31 - __clz(m);
			const int v = lv[hi];
			if (v != 1 && v != -1)
				break;
			t1++;
			m &= ~(1u << hi);
		}
	}
	int bits;
	if (chroma_dc) {
		bits = kCoeffTokenChromaDC[4 * n + t1] >> 8;
	} else {
		bits = 0;
#pragma unroll
		for (int tb = 0; tb < 4; tb++)
			bits = max(bits, (int)(kCoeffToken[tb * 68 + 4 * n + t1] >> 8));
	}
	if (n == 0)
		return bits;
	int suffix_len = (n > 10 && t1 < 3) ? 1 : 0;
	uint32_t m = mask;
	for (int k = 0; k < n; k++) {
		const int hi = 31 - __clz(m);
		m &= ~(1u << hi);
		if (k < t1) {
			bits += 1;
			continue;
		}
		const int level = lv[hi];
		const int a = level < 0 ? -level : level;
		int code = 2 * a - 2 + (level < 0);
		if (k == t1 && t1 < 3)
			code -= 2;
		if (suffix_len == 0)
			bits += code < 14 ? code + 1 : code < 30 ? 19 : 28;
		else if (code < (15 << suffix_len))
			bits += (code >> suffix_len) + 1 + suffix_len;
		else
			bits += 28;
		if (suffix_len == 0)
			suffix_len = 1;
		if (a > (3 << (suffix_len - 1)) && suffix_len < 6)
			suffix_len++;
	}
	const int last = 31 - __clz(mask);
	const int total_zeros = last + 1 - n;
	if (n < maxc)
		bits += (maxc == 4 ? kTotalZerosChromaDC[(n - 1) * 4 + total_zeros] : kTotalZeros[(n - 1) * 16 + total_zeros]) >> 8;
	int zeros_left = total_zeros;
	m = mask;
	for (int k = 0; k < n - 1 && zeros_left > 0; k++) {
		const int hi = 31 - __clz(m);
		m &= ~(1u << hi);
		const int next = 31 - __clz(m);
		const int run = hi - next - 1;
		bits += kRunBefore[((zeros_left > 7 ? 7 : zeros_left) - 1) * 16 + run] >> 8;
		zeros_left -= run;
	}
	return bits;
}

} // namespace cedar
