// cedarx_h264_encoder_amd/csrc/enc_headers.cpp
// The bytes the host writes itself and what follows from a picture's QP:
//   SPS / PPS / AUD (cedar.c:868-982; the slice headers are built per frame in enc_submit.cpp with the same bit writer)
//   quantiser / deblock thresholds of a QP (PARA1, cedar.c:1165-1168), the pictures that take I_PCM
//   the frame-level rate law (cedar_enc_config_ext.bitrate)
#include "enc_session.h"

namespace cedar {

// ---- ITU-T H.264 constants needed on the host (clauses 8.5.9, 8.7.2) ----
extern const uint16_t kQuantMF[6][3] = { // class 0 even/even, 1 odd/odd, 2 mixed
	{ 13107, 5243, 8066 }, { 11916, 4660, 7490 }, { 10082, 4194, 6554 },
	{ 9362, 3647, 5825 },  { 8192, 3355, 5243 },  { 7282, 2893, 4559 },
};
extern const uint8_t kDequantV[6][3] = {
	{ 10, 16, 13 }, { 11, 18, 14 }, { 13, 20, 16 }, { 14, 23, 18 }, { 16, 25, 20 }, { 18, 29, 23 },
};
extern const uint8_t kChromaQp[52] = {
	0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25,
	26, 27, 28, 29, 29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39
};
extern const uint8_t kAlpha[52] = {
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
	15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255
};
extern const uint8_t kBeta[52] = {
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
	6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18
};
extern const uint8_t kTc0[52][3] = {
	{ 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 },
	{ 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 }, { 0, 0, 0 },
	{ 0, 0, 0 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 1, 1 }, { 0, 1, 1 }, { 1, 1, 1 },
	{ 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 1 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 1, 2 }, { 1, 2, 3 },
	{ 1, 2, 3 }, { 2, 2, 3 }, { 2, 2, 4 }, { 2, 3, 4 }, { 2, 3, 4 }, { 3, 3, 5 }, { 3, 4, 6 }, { 3, 4, 6 },
	{ 4, 5, 7 }, { 4, 5, 8 }, { 4, 6, 9 }, { 5, 7, 10 }, { 6, 8, 11 }, { 6, 8, 13 }, { 7, 10, 14 }, { 8, 11, 16 },
	{ 9, 12, 18 }, { 10, 13, 20 }, { 11, 15, 23 }, { 13, 17, 25 }
};

// encoder policy: Lagrangian weight round(2^((qp-12)/6)), >= 1
int lambda_of_qp(int qp)
{
	static const uint8_t tab[52] = {
		1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4,
		5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25, 29, 32, 36, 40, 45, 51, 57, 64, 72, 81, 91
	};
	return tab[qp < 0 ? 0 : qp > 51 ? 51 : qp];
}

// NAL with 4-byte start code (cedar.c:868-881) and emulation prevention
void append_nal(std::vector<uint8_t> &out, int ref_idc, int type, const std::vector<uint8_t> &rbsp)
{
	out.push_back(0);
	out.push_back(0);
	out.push_back(0);
	out.push_back(1);
	out.push_back((uint8_t)(((ref_idc & 3) << 5) | (type & 0x1f)));
	int zeros = 0;
	for (uint8_t b : rbsp) {
		if (zeros == 2 && b <= 3) {
			out.push_back(3);
			zeros = 0;
		}
		out.push_back(b);
		zeros = b == 0 ? zeros + 1 : 0;
	}
}

// SPS (cedar.c:892-937) + PPS (cedar.c:939-982)
void build_sps_pps(cedar_enc *e)
{
	const struct cedar_enc_config &c = e->cfg;
	BitWriter sps;
	sps.put((uint32_t)c.profile, 8);
	sps.put(0, 8); // constraint flags byte = 0 (cedar.c:898-899)
	sps.put((uint32_t)c.level, 8);
	sps.ue(0); // seq_parameter_set_id
	sps.ue(0); // log2_max_frame_num_minus4
	sps.ue(2); // pic_order_cnt_type
	sps.ue(1); // max_num_ref_frames
	sps.put(0, 1);
	sps.ue((uint32_t)(c.dst_width / 16 - 1));
	sps.ue((uint32_t)(c.dst_height / 16 - 1));
	sps.put(1, 1); // frame_mbs_only_flag
	sps.put(0, 1); // direct_8x8_inference_flag
	if (c.dst_width != c.src_width || c.dst_height != c.src_height) {
		// crop units of 2 luma samples (the reference writes raw pixels, cedar.c:924-931)
		sps.put(1, 1);
		sps.ue(0);
		sps.ue((uint32_t)((c.dst_width - c.src_width) / 2));
		sps.ue(0);
		sps.ue((uint32_t)((c.dst_height - c.src_height) / 2));
	} else {
		sps.put(0, 1);
	}
	if (e->ext.fps_num > 0 && e->ext.fps_den > 0) {
		// H.264 E.1.1: timing + bitstream restriction, nothing else
		sps.put(1, 1); // vui_parameters_present_flag
		sps.put(0, 4); // aspect ratio, overscan, video signal type, chroma location: absent
		sps.put(1, 1); // timing_info_present_flag
		sps.put((uint32_t)e->ext.fps_den, 32);      // num_units_in_tick
		sps.put((uint32_t)e->ext.fps_num * 2u, 32); // time_scale: two ticks (fields) per frame
		sps.put(1, 1); // fixed_frame_rate_flag
		sps.put(0, 3); // nal_hrd, vcl_hrd, pic_struct: absent
		sps.put(1, 1); // bitstream_restriction_flag
		sps.put(1, 1); // motion_vectors_over_pic_boundaries_flag
		sps.ue(0);     // max_bytes_per_pic_denom: no limit
		sps.ue(0);     // max_bits_per_mb_denom: no limit
		sps.ue(7);     // log2_max_mv_length_horizontal: |mv| <= 16 pel = 64 quarter-pel
		sps.ue(7);     // log2_max_mv_length_vertical
		sps.ue(0);     // max_num_reorder_frames
		sps.ue(1);     // max_dec_frame_buffering
	} else {
		sps.put(0, 1); // vui_parameters_present_flag (cedar.c:933)
	}
	sps.trailing();
	BitWriter pps;
	pps.ue(0);
	pps.ue(0);
	pps.put(0, 1); // entropy_coding_mode_flag = CAVLC
	pps.put(0, 1);
	pps.ue(0);
	pps.ue(0);
	pps.ue(0);
	pps.put(0, 1);
	pps.put(0, 2);
	pps.se(c.qp - 26);
	pps.se(c.qp - 26);
	pps.se(e->chroma_qp_offset);
	pps.put(1, 1); // deblocking_filter_control_present_flag
	pps.put(e->ext.intra_refresh > 0, 1); // constrained_intra_pred_flag: refresh macroblocks must not lean on inter samples
	pps.put(0, 1);
	pps.trailing();
	e->sps_pps.clear();
	append_nal(e->sps_pps, 3, 7, sps.buf);
	append_nal(e->sps_pps, 3, 8, pps.buf);
	// access_unit_delimiter_rbsp (H.264 7.3.2.4): primary_pic_type 0 = I slices only, 1 = I and P
	for (int k = 0; k < 2; k++) {
		BitWriter aud;
		aud.put((uint32_t)k, 3);
		aud.trailing();
		e->aud_nal[k].clear();
		append_nal(e->aud_nal[k], 0, 9, aud.buf);
	}
}

// slice header bits (cedar.c:984-1030)
SliceHdr build_slice_header(const cedar_enc *e, bool idr, int frame_num, int idr_pic_id, int first_mb, int qp_delta)
{
	BitWriter b;
	b.ue((uint32_t)first_mb);   // reference: always 0 (cedar.c:992-993)
	b.ue(idr ? 2 : 0);          // slice_type
	b.ue(0);                    // pic_parameter_set_id
	b.put((uint32_t)(frame_num & 0x0f), 4);
	if (idr) {
		b.ue((uint32_t)idr_pic_id);
		b.put(0, 1);
		b.put(0, 1);
	} else {
		b.put(0, 1);
		b.put(0, 1);
		b.put(0, 1);
	}
	b.se(qp_delta);                           // slice_qp_delta: 0 at fixed QP (cedar.c:1021-1022)
	b.ue(e->ext.slice_count > 1 ? 2 : 0);     // disable_deblocking_filter_idc
	b.se(0);
	b.se(0);
	SliceHdr h{};
	h.nbits = (int32_t)b.nbits;
	uint64_t v = 0;
	for (uint64_t i = 0; i < b.nbits; i++)
		v = (v << 1) | ((b.buf[i >> 3] >> (7 - (i & 7))) & 1);
	h.bits = v;
	return h;
}

// everything of Params that follows from the picture's QP (PARA1 in the reference, cedar.c:1165-1168:
// one fixed QP; with rate control one per picture)
void set_qp_params(Params &P, int qp, int chroma_qp_offset)
{
	P.qp = qp;
	int qi = qp + chroma_qp_offset;
	P.qpc = kChromaQp[qi < 0 ? 0 : qi > 51 ? 51 : qi];
	P.lambda = lambda_of_qp(qp);
	for (int k = 0; k < 3; k++) {
		P.mf[k] = kQuantMF[P.qp % 6][k];
		P.dv[k] = kDequantV[P.qp % 6][k];
		P.mfc[k] = kQuantMF[P.qpc % 6][k];
		P.dvc[k] = kDequantV[P.qpc % 6][k];
		P.tc0_y[k] = kTc0[P.qp][k];
		P.tc0_c[k] = kTc0[P.qpc][k];
	}
	P.qbits = 15 + P.qp / 6;
	P.qshift = P.qp / 6;
	P.qbits_c = 15 + P.qpc / 6;
	P.qshift_c = P.qpc / 6;
	P.alpha_y = kAlpha[P.qp];
	P.beta_y = kBeta[P.qp];
	P.alpha_c = kAlpha[P.qpc];
	P.beta_c = kBeta[P.qpc];
}

// I_PCM (cedar_enc_config_ext.pcm): the pictures that take it are those at whose QP the edges of an I_PCM
// macroblock are not filtered at all. The deblocking filter takes its QP_Y as 0 (8.7.2.2) and filters an edge at
// the mean of the two macroblocks' QP_Y resp. QP_C; alpha is 0 up to index 15.
bool picture_takes_pcm(int qp, int chroma_qp_offset)
{
	auto qpc = [&](int q) { const int qi = q + chroma_qp_offset; return (int)kChromaQp[qi < 0 ? 0 : qi > 51 ? 51 : qi]; };
	return kAlpha[(qp + 1) >> 1] == 0 && kAlpha[(qpc(qp) + qpc(0) + 1) >> 1] == 0;
}

// ---- frame-level rate control (cedar_enc_config_ext.bitrate; the tests hold a scalar
// twin of this law against it). Integer only: frame n looks at the sizes of frames <= n - 1 - RC_LAG (collected whatever
// the caller's pipelining: a session keeps at most RC_LAG frames in flight), over the last keyint of
// them; each is scaled to the configured QP by "6 QP steps halve the bits" (Q16); the new QP is the
// lowest one at which that model predicts no more than the target.
constexpr int RC_LAG = CEDAR_ENC_PIPELINE_DEPTH;

int64_t rc_pow2_6_q16(int d) // 2^(d/6) in Q16
{
	static const int64_t t[6] = { 65536, 73562, 82570, 92682, 104032, 116772 };
	const int q = d >= 0 ? d / 6 : -((-d + 5) / 6), r = d - 6 * q;
	return q >= 0 ? t[r] << q : t[r] >> -q;
}

int rc_pick_qp(int qp0, int64_t target, const int64_t *bits, const int *qps, long n, int keyint)
{
	const long known = n - RC_LAG;
	if (known <= 0 || target <= 0)
		return qp0;
	const long w = known < keyint ? known : keyint;
	int64_t sum = 0;
	for (long i = known - w; i < known; i++)
		sum += (bits[i & 63] * rc_pow2_6_q16(qps[i & 63] - qp0)) >> 16;
	const int64_t avg = sum / w;
	int d = -46;
	while (d < 46 && rc_pow2_6_q16(d) * target < (avg << 16))
		d++;
	const int qp = qp0 + d;
	return qp < 1 ? 1 : qp > 47 ? 47 : qp;
}

// The law of a session that codes G closed GOPs side by side: nothing inside a group can steer it (the later GOPs of a
// group are coded beside the earlier ones), so there is ONE QP per group, from the mean bits per frame -- scaled to the
// configured QP as above -- of the group 1 + ceil(16 / keyint) groups back: the nearest group all of whose lock-step steps
// have been collected when the first step of this one is submitted, whatever the caller's pipelining (16 steps in flight at
// most). DESIGN.md 3.2; the scalar checker of the tests runs the same law serially.
int rc_group_lag(int keyint) { return 1 + (CEDAR_ENC_PIPELINE_DEPTH + keyint - 1) / keyint; }
int rc_pick_qp_group(int qp0, int64_t target, const int64_t *sum, const int *cnt, const long *id1, long m, int keyint)
{
	const long r = m - rc_group_lag(keyint);
	if (r < 0 || target <= 0 || id1[r & 31] != r + 1 || cnt[r & 31] <= 0)
		return qp0;
	const int64_t avg = sum[r & 31] / cnt[r & 31];
	int d = -46;
	while (d < 46 && rc_pow2_6_q16(d) * target < (avg << 16))
		d++;
	const int qp = qp0 + d;
	return qp < 1 ? 1 : qp > 47 ? 47 : qp;
}

} // namespace cedar
