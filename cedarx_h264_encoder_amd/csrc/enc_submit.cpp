// cedarx_h264_encoder_amd/csrc/enc_submit.cpp
// CEDAR_IOCTL_ENCODE (cedar.c:1032-1209) as submit + collect: "program + trigger" becomes the launches of a frame on the
// session's streams, "wait for the IRQ" the completion words a collect polls. Also what the reference has no word for:
// frames in flight, a failed frame dropped and the frames behind it replayed, the held-back tail of a pipelined picture.
#include "enc_session.h"

namespace cedar {

static double now_ms()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static void tm_begin(cedar_enc *e, InFlight &f, int stage, hipStream_t s)
{
	if (e->profiling && ((e->profile_mask >> stage) & 1u) && e->frame_count % e->profile_every == 0) {
		hipEventRecord(f.tm.ev[stage][0], s);
		f.tm.used[stage] = true;
	}
}
static void tm_end(cedar_enc *e, InFlight &f, int stage, hipStream_t s)
{
	if (f.tm.used[stage])
		hipEventRecord(f.tm.ev[stage][1], s);
}

static Params cohort_params(const cedar_enc *e, const Params &P, int c)
{
	const int C = e->cohorts;
	if (C == 1)
		return P;
	const int S = P.n_streams, s0 = S * c / C, n = S * (c + 1) / C - s0;
	const size_t u0 = (size_t)s0 * P.slice_num, m0 = (size_t)s0 * P.n_mb;
	Params Q = P;
	Q.n_streams = n;
	Q.dummy_mb = (int64_t)((size_t)S * P.n_mb - m0);
	if (Q.in_luma) {
		Q.in_luma += s0;
		Q.in_chroma += s0;
	} else {
		for (int s = 0; s < IN_K_STREAMS; s++) {
			Q.in_k[s] = s0 + s < IN_K_STREAMS ? P.in_k[s0 + s] : nullptr;
			Q.in_k[IN_K_STREAMS + s] = s0 + s < IN_K_STREAMS ? P.in_k[IN_K_STREAMS + s0 + s] : nullptr;
		}
	}
	Q.src_y += (size_t)s0 * P.plane_y_stride;
	Q.src_uv += (size_t)s0 * P.plane_uv_stride;
	Q.rec_y += (size_t)s0 * P.plane_y_stride;
	Q.rec_uv += (size_t)s0 * P.plane_uv_stride;
	Q.ref_y += (size_t)s0 * P.plane_y_stride;
	Q.ref_uv += (size_t)s0 * P.plane_uv_stride;
	Q.mbi += m0;
	if (Q.qp_s)
		Q.qp_s += s0;
	if (Q.qp_eff) {
		Q.qp_eff += m0;
		Q.qp_t += m0;
	}
	Q.i4modes += m0 * 16;
	Q.nnz += m0 * NNZ_PER_MB;
	Q.coef += m0 * COEF_PER_MB;
	Q.slot += m0 * SLOT_WORDS;
	Q.mb_bits += m0;
	Q.skip += m0;
	Q.bit_off += u0 * (P.unit_mb_cap + 2);
	Q.run_before += u0 * (P.unit_mb_cap + 2);
	Q.unit_bits += u0;
	if (Q.rbsp) {
		Q.rbsp += u0 * P.rbsp_cap;
		Q.esc_cnt += u0 * P.esc_chunks_cap;
	}
	if (Q.ent_rec) {
		Q.ent_rec += u0 * (P.unit_mb_cap + 2);
		Q.pk_first += u0 * P.pk_chunks_cap;
		Q.pk_agg += u0 * P.pk_chunks_cap;
		Q.pk_unit += u0;
	}
	Q.nal_off += u0;
	Q.nal_size += u0;
	Q.hdr += u0;
	Q.prefix_len += s0;
	Q.stream_size += (size_t)c * (S + 4);
	if (Q.out)
		Q.out += (size_t)s0 * e->stream_out_cap;
	Q.out_cap = (int32_t)((size_t)n * e->stream_out_cap);
	Q.wf_sync += (size_t)c * 3 * WF_SYNC_WORDS;
	Q.wf_hbuf += u0 * e->hb_slots * P.mb_w * WF_HBUF_GRANULES;
	if (Q.db_hbuf)
		Q.db_hbuf += u0 * e->db_bands * P.mb_w * DBR_HB_GRANULES;
	Q.wf_hbuf_intra += u0 * e->hb_slots * P.mb_w * WF_HBUF_GRANULES;
	if (Q.ir_hbuf)
		Q.ir_hbuf += u0 * 2 * e->db_bands * 4 * P.mb_w;
	return Q;
}


// The stripes [k0, k1) of a pipelined P picture's analysis: a gate (publish how far this picture's
// analysis is, wait for the previous picture's deblocking) and the search of the stripe, each.
static int emit_stripes(cedar_enc *e, Params &Pc, hipStream_t s_ana, int k0, int k1)
{
	const int S = e->ext.n_streams;
	for (int k = k0; k < k1; k++) {
		Pc.stripe_s0 = e->stripe_s[k];
		Pc.stripe_s1 = e->stripe_s[k + 1];
		Pc.stripe_tab = e->d_stripe_tab + (size_t)k * (Pc.row_num + 1);
		// tail_gate: the stripe before this one has waited for this one's gate (one of its waves, k_inter), and this launch's
		// first wave publishes that stripe's end: no k_gate in between. Only between two stripes that both have macroblocks.
		const bool gated_by_prev = e->tail_gate && k > 0 && e->stripe_n[k] > 0 && e->stripe_n[k - 1] > 0;
		const bool gates_next = e->tail_gate && k + 1 < e->inter_bands && e->stripe_n[k] > 0 && e->stripe_n[k + 1] > 0;
		if (!gated_by_prev && Launch::gate(Pc, s_ana, k > 0 ? Pc.stripe_s0 : -1, Pc.stripe_s1 + 4))
			return fail(e, -EIO, "k_gate launch failed");
		Pc.gate_publish = gated_by_prev ? Pc.stripe_s0 : -1;
		Pc.gate_next_need = gates_next ? e->stripe_s[k + 2] + 4 : 0;
		if (e->stripe_n[k] > 0 && Launch::inter(Pc, s_ana, e->stripe_n[k] * S))
			return fail(e, -EIO, "k_inter launch failed");
		Pc.gate_publish = -1;
		Pc.gate_next_need = 0;
	}
	return 0;
}

// What follows the last stripe of a pipelined P picture: the gate that publishes "all of it", the
// analysis event, the entropy launches behind it.
static int emit_tail(cedar_enc *e, Params &Pc, hipStream_t s_ana, hipStream_t s_ent, int slot, int ei)
{
	InFlight &f = e->fl[slot];
	if (Launch::gate(Pc, s_ana, PROG_ALL, 0))
		return fail(e, -EIO, "k_gate launch failed");
	Pc.stripe_s0 = Pc.stripe_s1 = 0;
	tm_end(e, f, CEDAR_ENC_STAGE_INTER, s_ana);
	HIP_TRY(e, hipEventRecord(e->ev_analysis[ei], s_ana));
	HIP_TRY(e, hipStreamWaitEvent(s_ent, e->ev_analysis[ei], 0));
	tm_begin(e, f, CEDAR_ENC_STAGE_ENTROPY, s_ent);
	if (Pc.roi_off && Launch::qp_eff(Pc, s_ent))
		return fail(e, -EIO, "k_qp_eff launch failed");
	Pc.done_flag = f.d_done;
	Pc.done_tag = f.done_tag;
	if (Launch::entropy_out(Pc, s_ent, e->dv_out[f.host_out], e->dv_sizes[slot]))
		return fail(e, -EIO, "entropy launch failed");
	tm_end(e, f, CEDAR_ENC_STAGE_ENTROPY, s_ent);
	return 0;
}

// the held-back part of the newest pipelined P picture, now (nothing will be enqueued in front of it)
int flush_deferred(cedar_enc *e)
{
	cedar_enc::DeferredTail &dt = e->deferred;
	if (!dt.active)
		return 0;
	dt.active = false;
	if (e->flusher)
		e->flusher->armed = false;
	int rc = emit_stripes(e, dt.Pc, dt.s_ana, dt.next, e->inter_bands);
	if (!rc)
		rc = emit_tail(e, dt.Pc, dt.s_ana, dt.s_ent, dt.slot, dt.ei);
	return rc;
}

// the session's flush thread (see Flusher): sleeps until a tail is held back, then until its deadline
void flusher_main(cedar_enc *e)
{
	Flusher &F = *e->flusher;
	std::unique_lock<std::mutex> lk(F.mu);
	while (!F.stop) {
		if (!F.armed) {
			F.cv.wait(lk);
			continue;
		}
		if (std::chrono::steady_clock::now() < F.deadline) {
			F.cv.wait_until(lk, F.deadline); // (a submit moves the deadline and wakes us: look again)
			continue;
		}
		F.armed = false;
		if (e->deferred.active) {
			hipSetDevice(e->device);
			const int rc = flush_deferred(e);
			if (rc && !F.rc)
				F.rc = rc;
			F.flushes++;
		}
	}
}

// a tail has just been held back: it goes out on its own after Flusher::us
static void flusher_arm(cedar_enc *e)
{
	if (!e->flusher)
		return;
	e->flusher->deadline = std::chrono::steady_clock::now() + std::chrono::microseconds(e->flusher->us);
	e->flusher->armed = true;
	e->flusher->cv.notify_one();
}

void flusher_stop(cedar_enc *e)
{
	if (!e->flusher)
		return;
	{
		std::lock_guard<std::mutex> lk(e->flusher->mu);
		e->flusher->stop = true;
		e->flusher->cv.notify_one();
	}
	if (e->flusher->th.joinable())
		e->flusher->th.join();
	delete e->flusher;
	e->flusher = nullptr;
}

// The input stage of a submit: H2D of the pinned buffers unless the caller supplied device frames (a replayed frame: its data
// moves from the slot it was first submitted to), where the kernels find stream s of this frame (kernel arguments up to 8
// streams, a pointer table beyond), and whether the frame can be read in place (Params::src_direct). in_work: something of
// this frame was enqueued on s_in.
static int submit_input(cedar_enc *e, InFlight &f, Params &P, const ReplayIn *replay, int ds, hipStream_t s_in, bool &in_work, bool &any_host)
{
	const int S = e->ext.n_streams;
	any_host = in_work = false;
	const size_t in_y_slot = (size_t)ds * e->cfg.input_luma_size * S, in_uv_slot = (size_t)ds * e->cfg.input_chroma_size * S;
	f.in_own.assign((size_t)S, 0);
	for (int s = 0; s < S; s++) {
		// a stream without a device frame of its own points at the session's device copy (slot 0 is the mark)
		const uint8_t *own = e->d_in_y + (size_t)s * e->cfg.input_luma_size;
		f.in_own[s] = replay ? replay->own[s] : (uint8_t)(e->h_in_ptrs[s] == own);
		if (!f.in_own[s])
			continue;
		uint8_t *dy = e->d_in_y + in_y_slot + (size_t)s * e->cfg.input_luma_size;
		uint8_t *duv = e->d_in_uv + in_uv_slot + (size_t)s * e->cfg.input_chroma_size;
		if (replay) {
			// the frame's data sits in the device copy of the slot it was first submitted to: move it to
			// this slot's, so that "a frame in flight owns the copy of its slot" holds again
			if (replay->in[s] != dy) {
				in_work = true;
				HIP_TRY(e, hipMemcpyAsync(dy, replay->in[s], (size_t)P.width * P.height, hipMemcpyDeviceToDevice, s_in));
				HIP_TRY(e, hipMemcpyAsync(duv, replay->in[S + s], (size_t)P.width * P.height / 2, hipMemcpyDeviceToDevice, s_in));
			}
			continue;
		}
		any_host = in_work = true;
		HIP_TRY(e, hipMemcpyAsync(dy, e->h_luma + (size_t)s * e->cfg.input_luma_size,
					  (size_t)P.width * P.height, hipMemcpyHostToDevice, s_in));
		HIP_TRY(e, hipMemcpyAsync(duv, e->h_chroma + (size_t)s * e->cfg.input_chroma_size,
					  (size_t)P.width * P.height / 2, hipMemcpyHostToDevice, s_in));
	}
	auto in_ptr = [&](int s, bool chroma) -> const uint8_t * { // where k_input finds stream s of this frame
		if (!f.in_own[s])
			return replay ? replay->in[(chroma ? S : 0) + s] : e->h_in_ptrs[(chroma ? S : 0) + s];
		return chroma ? e->d_in_uv + in_uv_slot + (size_t)s * e->cfg.input_chroma_size
			      : e->d_in_y + in_y_slot + (size_t)s * e->cfg.input_luma_size;
	};
	f.in_used.resize((size_t)2 * S);
	for (int s = 0; s < S; s++) {
		f.in_used[s] = in_ptr(s, false);
		f.in_used[S + s] = in_ptr(s, true);
	}
	// (a plain session counts its frames itself; a replayed frame comes with its index, which recover()
	// has moved up by the frame that was dropped; GOP lanes keep the frame they were given)
	if (replay)
		f.lane_gidx = replay->gidx;
	else if (!e->mux)
		f.lane_gidx.assign((size_t)S, (long)e->ext.first_frame_index + e->frame_count);
	else
		f.lane_gidx = e->lane_gidx;
	if (replay)
		e->in_tab_valid[ds] = false; // the table of this slot, if any, describes another frame
	if (S <= IN_K_STREAMS) { // frame pointers ride in the kernel arguments: no table upload
		P.in_luma = P.in_chroma = nullptr;
		for (int s = 0; s < S; s++) {
			P.in_k[s] = in_ptr(s, false);
			P.in_k[IN_K_STREAMS + s] = in_ptr(s, true);
		}
	} else if (!e->in_tab_valid[ds]) {
		// many streams: a pointer table in device memory per slot (the device copies of host frames
		// rotate with the slot), uploaded when the caller changed a pointer
		std::vector<const uint8_t *> &t = e->in_tab[ds];
		t.resize((size_t)2 * S);
		for (int s = 0; s < S; s++) {
			t[s] = in_ptr(s, false);
			t[S + s] = in_ptr(s, true);
		}
		in_work = true;
		HIP_TRY(e, hipMemcpyAsync((void *)(e->d_in_ptrs + (size_t)ds * 2 * S), t.data(), sizeof(void *) * 2 * S,
					  hipMemcpyHostToDevice, s_in));
		e->in_tab_valid[ds] = !replay;
	}
	if (S > IN_K_STREAMS) {
		P.in_luma = (const uint8_t *const *)(e->d_in_ptrs + (size_t)ds * 2 * S);
		P.in_chroma = P.in_luma + S;
	}
	if (any_host)
		HIP_TRY(e, hipEventRecord(e->ev_host_in, s_in));
	// No padded copy of the frame (no k_input: -6.3 MB of traffic and a launch per 1080p frame) where the caller's packed
	// frame can serve as the source planes: a width that is a multiple of 16 (no column to replicate; the rows below the
	// picture are the last row, which the readers take by clamping) and 16-byte aligned frames (they load 16 bytes a lane).
	{
		bool direct = e->src_direct_ok;
		for (int s = 0; s < 2 * S && direct; s++)
			direct = ((uintptr_t)f.in_used[s] & 15u) == 0;
		P.src_direct = direct ? 1 : 0;
		e->last_src_direct = direct;
		P.src_last_y = direct ? P.height - 1 : P.ch - 1;
		P.src_last_c = direct ? P.height / 2 - 1 : P.ch / 2 - 1;
	}
	return 0;
}

static void build_frame_headers(cedar_enc *e, InFlight &f, const Params &P, bool idr)
{
	const int S = e->ext.n_streams;
	// Headers per lane: idr_pic_id and "parameter sets in front of frame 0" follow the GLOBAL frame
	// index (the reference counts the frames of its one device, cedar.c:1001, :1058-1061; a shard or a
	// GOP lane starts somewhere in the stream). frame_num restarts with every IDR picture (cedar.c:1001).
	f.aud = e->ext.aud && e->ext.slice_first == 0 ? (idr ? 1 : 2) : 0; // the first slice's session opens the access unit
	f.with_ps.assign((size_t)S, 0);
	for (int s = 0; s < S; s++) {
		const long gidx = f.lane_gidx[s] < 0 ? 0 : f.lane_gidx[s];
		const long gop = gidx / e->cfg.keyframe_interval;
		for (int k = 0; k < e->ext.slice_num; k++) {
			int r0, r1;
			slice_rows(P.mb_h, P.slice_count, e->ext.slice_first + k, &r0, &r1);
			// (scene cuts put IDR pictures anywhere: two in a row must still differ in idr_pic_id, so it follows their count)
			f.h_hdr[s * e->ext.slice_num + k] = build_slice_header(e, idr, e->frame_p_count, e->ext.scene_cut > 0 ? (int)(e->idr_count & 1u) : (int)(gop & 1), r0 * P.mb_w,
									       (e->rc_multi ? (int)f.h_qp_s[s] : P.qp) - e->cfg.qp);
		}
		// cedar.c:1058-1061: SPS + PPS in front of the stream's first picture (the IDR picture of GOP 0)
		f.with_ps[s] = idr && ((e->ext.scene_cut > 0 ? gidx == 0 : gop == 0) || e->ext.repeat_headers) && e->ext.slice_first == 0;
		f.h_prefix[s] = (f.with_ps[s] ? (uint32_t)e->sps_pps.size() : 0) + (f.aud ? (uint32_t)e->aud_nal[f.aud - 1].size() : 0);
	}
}

// What one submit works with: the frame, its launch parameters, the streams that carry it.
struct FrameCtx {
	cedar_enc *e;
	InFlight *f;
	Params P;
	int S, slot, ds, prev, C; // streams, ring slot, device slot, device slot of the frame before, cohorts
	bool idr, pipe;           // an IDR picture; consecutive frames overlap (cedar_enc::row_pipe)
	bool ahead = false, db_late = false;        // see frame_streams
	bool self_zero = false, self_reset = false; // see submit_frame
	bool in_work = false;                       // something of this frame was enqueued on s_in
	int lane = 0;                               // entropy lane
	hipStream_t s_db = nullptr, s_ana = nullptr, s_in = nullptr, s_ent = nullptr;
};

// The frame's launch parameters: its QP (the rate law, a replay's bump), the buffers of its device slot, the
// reference ping-pong (cedar.c:1193-1201), the progress words and tags of pipelined frames.
static void frame_params(cedar_enc *e, InFlight &f, Params &P, const ReplayIn *replay, bool idr, int ds, uint32_t seq)
{
	const int S = e->ext.n_streams;
	P.is_idr = idr;
	P.refresh_t = e->frame_p_count; // P pictures since the IDR (cyclic intra refresh)
	f.qp_bump = replay ? replay->qp_bump : 0;
	if (e->rc_target > 0 || f.qp_bump)
		set_qp_params(P, (e->rc_lanes	   ? rc_pick_qp_group(e->cfg.qp, e->rc_target, e->grp_sum, e->grp_n, e->grp_id1,
								      e->frame_count / e->cfg.keyframe_interval, e->cfg.keyframe_interval)
				  : e->rc_target > 0 ? rc_pick_qp(e->cfg.qp, e->rc_target, e->rc_bits, e->rc_qp, e->frame_count, e->cfg.keyframe_interval)
						     : e->cfg.qp) + f.qp_bump,
			      e->chroma_qp_offset);
	f.qp = P.qp;
	P.pcm = e->pcm_session && picture_takes_pcm(P.qp, e->chroma_qp_offset) ? 1 : 0;
	P.cut_votes = e->ext.scene_cut > 0 && !idr ? e->d_cut_votes[ds] : nullptr; // scene cuts: P pictures are asked
	if (e->rc_multi) { // the law per stream; the launch's own QP (Params::qp) is not used by the per-macroblock forms
		for (int s = 0; s < S; s++)
			f.h_qp_s[s] = (uint8_t)rc_pick_qp(e->cfg.qp, e->rc_target, &e->rc_bits_s[(size_t)s * 64], &e->rc_qp_s[(size_t)s * 64],
							 e->frame_count, e->cfg.keyframe_interval);
		P.qp_s = f.dv_qp_s;
		f.qp = f.h_qp_s[0];
	}
	if (e->fault_at && seq != e->fault_at) // tests: lose the hand-off of one chosen submit only
		P.fault_inject = 0;
	P.rec_y = e->d_rec_y[e->cur];
	P.rec_uv = e->d_rec_uv[e->cur];
	const int ref = (e->cur + e->nrec - 1) % e->nrec;
	P.ref_y = e->d_rec_y[ref];
	P.ref_uv = e->d_rec_uv[ref];
	P.src_y = e->d_src_y[ds];
	P.src_uv = e->d_src_uv[ds];
	P.mbi = e->d_mbi[ds];
	P.qp_eff = e->d_qp_eff[ds];
	P.qp_t = e->d_qp_t[ds];
	P.i4modes = e->d_i4[ds];
	P.nnz = e->d_nnz[ds];
	P.coef = e->d_coef[ds];
	P.wf_sync = e->d_wf_sync[ds];
	P.wf_hbuf = e->d_wf_hbuf[ds];
	P.db_hbuf = e->d_db_hbuf[ds];
	P.ir_hbuf = e->d_ir_hbuf[ds];
	P.db_tag = seq;
	P.wf_hbuf_intra = e->d_wf_hbuf[ds] + (size_t)S * e->ext.slice_num * e->hb_slots * P.mb_w * WF_HBUF_GRANULES;
	P.out = e->d_out[ds];
	const int prev = e->prev_slot;
	if (e->row_pipe) {
		P.prog_inter = e->d_prog[ds];
		P.prog_db = e->d_prog[ds] + 16;
		P.prog_intra = e->d_prog[ds] + 16 + 2 * (size_t)S * e->ext.slice_num * PROG_BANDS;
		P.prog_intra_c = e->ir_rows ? e->d_prog[ds] + 16 + 3 * (size_t)S * e->ext.slice_num * PROG_BANDS : nullptr;
		P.prev_db = prev >= 0 ? e->d_prog[prev] + 16 : nullptr;
		P.prog_tag = seq << 16;
		P.prev_tag = prev >= 0 ? e->slot_tag[prev] : 0;
		e->slot_tag[ds] = P.prog_tag;
	}
	P.nal_ref_idc = idr ? 3 : 2; // cedar.c:987-990
	P.nal_type = idr ? 5 : 1;
}

// Which streams carry this frame. With look-ahead an IDR picture has streams of its own and
// waits for nothing but its input: it runs beside the P pictures submitted before it.
static void frame_streams(FrameCtx &x)
{
	cedar_enc *e = x.e;
	x.ahead = e->lookahead && x.idr;
	// (idr_deblock_inline: the IDR picture's deblocking follows its analysis on the same stream
	// instead of running beside it on a stream of its own, resident and polling)
	x.db_late = x.ahead && e->idr_deblock_inline;
	hipStream_t s_idr = x.ahead && e->s_idr2 && (e->idr_count & 1u) ? e->s_idr2 : e->s_idr; // IDR pictures take turns
	x.s_db = x.db_late ? s_idr : x.ahead ? e->s_db_idr : e->s_db[e->frame_count % e->db_streams];
	x.s_ana = x.ahead ? s_idr : e->s_ana[e->frame_count % e->an_streams]; // frames take turns: each stream's gates wait for one frame only
	x.s_in = e->lookahead ? e->s_in : x.s_ana; // input of all frames, in order, behind no gate
	// entropy lanes: frames alternate between two streams with scratch buffers of their own
	x.lane = e->ent_lanes > 1 && e->s_ent2 ? e->frame_count % e->ent_lanes : 0;
	x.s_ent = x.lane ? e->s_ent2 : e->s_ent;
}

// the scratch of the frame's entropy lane, and how many workgroups k_pack_out gets
static void entropy_lane_params(const cedar_enc *e, Params &P, int lane, bool idr)
{
	P.slot = e->d_slot[lane];
	P.mb_bits = e->d_mb_bits[lane];
	P.skip = e->d_skip[lane];
	P.bit_off = e->d_bit_off[lane];
	P.run_before = e->d_run[lane];
	P.ent_rec = e->d_ent_rec[lane];
	{
		P.pk_deliver = getenv("CEDAR_PK_DELIVER") ? atoi(getenv("CEDAR_PK_DELIVER")) : 0;
		const char *g = getenv("CEDAR_PK_GRID"); // (development: workgroups per unit, fixed)
		const int seen = e->pk_seen[idr ? 1 : 0], unit_mbs = P.unit_mb_cap;
		P.pk_launch_chunks = g ? atoi(g) : seen ? 2 * seen + 2 : idr ? unit_mbs / 12 + 8 : unit_mbs / 96 + 4; // (80 / 10 bytes a macroblock)
	}
	P.pk_first = e->d_pk_first[lane];
	P.pk_agg = e->d_pk_agg[lane];
	P.pk_unit = e->d_pk_unit[lane];
	P.unit_bits = e->d_unit_bits[lane];
	P.rbsp = e->d_rbsp[lane];
	P.esc_cnt = e->d_esc[lane];
	P.nal_off = e->d_nal_off[lane];
	P.nal_size = e->d_nal_size[lane];
	P.stream_size = e->d_stream_size[lane];
}

// ---- in-loop deblocking of cohort c: produces the next frame's reference ----
// Row-pipelined it starts at once and polls the analysis progress itself; otherwise it
// waits for the whole analysis.
static int enqueue_deblock(FrameCtx &x, Params &Pc, int c, hipStream_t s_dbc)
{
	cedar_enc *e = x.e;
	InFlight &f = *x.f;
	const bool pipe = x.pipe;
	const int C = x.C, ei = x.ds * MAX_COHORTS + c;
	// No event here: a cross-stream wait resolves to a later point of s_main than the record.
	// The kernel needs nothing of this frame's s_main work up front: its sync words are zeroed
	// on its own stream, the progress words need no zeroing (frame tags).
	if (!pipe)
		HIP_TRY(e, hipStreamWaitEvent(s_dbc, e->ev_analysis[ei], 0));
	if (x.db_late) // same stream: everything is there when the kernel starts, nothing to poll
		Pc.prog_intra = Pc.prog_intra_c = nullptr;
	// Pipelined: ONE packet on the deblock stream per frame -- every packet queued behind a
	// resident kernel costs the hardware pipe ~20 us. The kernel's last workgroup zeroes the
	// sync words for the slot's next frame and raises the frame's done flag itself.
	// Frames back to back do the same (a memset or a 1-thread kernel that meets a chip full of k_inter
	// workgroups waits for a CU like everybody else: 32 x 1080p showed the wf_sync memset take 377 us
	// in front of a deblock kernel). Cohorts follow each other on one stream: the last one's flag says all.
	Pc.db_self_reset = pipe || x.self_zero ? 1 : 0;
	Pc.done_flag = pipe || (x.self_reset && c == C - 1) ? f.d_done + 1 : nullptr;
	Pc.done_tag = f.done_tag;
	if (!Pc.db_self_reset && (e->wf_split > 1 || e->db_rows))
		HIP_TRY(e, hipMemsetAsync(Pc.wf_sync + WF_SYNC_WORDS, 0,
					  sizeof(uint32_t) * (WF_SYNC_WORDS + 16 + (size_t)Pc.n_streams * e->ext.slice_num *
								      (e->db_rows ? e->db_bands : WF_MAX_SPLIT)), s_dbc));
	if (c == 0)
		tm_begin(e, f, CEDAR_ENC_STAGE_DEBLOCK, s_dbc);
	// luma and chroma are independent wavefronts: one launch, the plane picked per workgroup
	// (few units), or both planes in every wave pair (many units: CUs are the scarce thing)
	if (Launch::deblock(Pc, s_dbc, e->split_planes ? 4 : 3))
		return fail(e, -EIO, "k_deblock launch failed");
	if (c == C - 1)
		tm_end(e, f, CEDAR_ENC_STAGE_DEBLOCK, s_dbc);
	Pc.db_self_reset = 0;
	if (!pipe)
		HIP_TRY(e, hipEventRecord(e->ev_deblocked[ei], s_dbc));
	if (!pipe && c == C - 1) {
		// the frame is complete when deblocking and entropy coding are (collect waits for both)
		for (int c2 = 0; c2 < c; c2++) // (cohorts on streams of their own)
			if ((!x.ahead && e->s_db_c[c2] ? e->s_db_c[c2] : x.s_db) != s_dbc)
				HIP_TRY(e, hipStreamWaitEvent(s_dbc, e->ev_deblocked[x.ds * MAX_COHORTS + c2], 0));
		HIP_TRY(e, hipEventRecord(f.done_main, s_dbc));
		if (!x.self_reset && Launch::flag(s_dbc, f.d_done + 1, f.done_tag))
			return fail(e, -EIO, "k_flag launch failed");
	}
	return 0;
}

// ---- analysis + reconstruction of cohort c ----
// What this frame overwrites or reads of its predecessors: rec[cur] was last touched by frames
// that have been collected (the buffers rotate through depth + 1); a P picture reads the
// reconstruction frame n-1 is deblocking. Row-pipelined, a P picture follows that deblocking
// band by band (k_gate); otherwise it waits for the whole frame. An IDR picture waits for nothing.
// tail_done: a pipelined P picture -- emit_tail has (or will have) done the rest of the frame.
static int enqueue_analysis(FrameCtx &x, Params &Pc, int c, bool &tail_done)
{
	cedar_enc *e = x.e;
	InFlight &f = *x.f;
	const bool pipe = x.pipe;
	const int C = x.C, ei = x.ds * MAX_COHORTS + c, eprev = x.prev * MAX_COHORTS + c;
	hipStream_t s_ana = x.s_ana;
	if (x.prev >= 0 && !x.idr && !pipe)
		HIP_TRY(e, hipStreamWaitEvent(s_ana, e->ev_deblocked[eprev], 0));
	if (x.idr) {
		if (c == 0)
			tm_begin(e, f, CEDAR_ENC_STAGE_INTRA_DECIDE, s_ana);
		if (Launch::intra_decide(Pc, s_ana))
			return fail(e, -EIO, "k_intra_decide launch failed");
		if (C == 1)
			tm_end(e, f, CEDAR_ENC_STAGE_INTRA_DECIDE, s_ana);
		if (C == 1)
			tm_begin(e, f, CEDAR_ENC_STAGE_INTRA_RECON, s_ana);
		if (Launch::intra_recon(Pc, s_ana))
			return fail(e, -EIO, "k_intra_recon launch failed");
		if (C == 1)
			tm_end(e, f, CEDAR_ENC_STAGE_INTRA_RECON, s_ana);
		else if (c == C - 1) // cohorts: both intra stages of all cohorts, under the first stage's name
			tm_end(e, f, CEDAR_ENC_STAGE_INTRA_DECIDE, s_ana);
		if (pipe && Launch::gate(Pc, s_ana, PROG_ALL, 0))
			return fail(e, -EIO, "k_gate launch failed");
		return 0;
	}
	if (c == 0)
		tm_begin(e, f, CEDAR_ENC_STAGE_INTER, s_ana);
	if (pipe) {
		// diagonal stripes s = x + y: the order in which the deblock wavefront of the
		// previous frame releases the reference and in which this frame's own can follow.
		// The previous picture's held-back stripes go in between this picture's first ones
		// (cedar_enc::DeferredTail), this picture's last ones are held back in turn.
		const int nb = e->inter_bands, kd = e->defer_from;
		cedar_enc::DeferredTail &dt = e->deferred;
		int own = 0, rc_s = 0;
		if (dt.active) {
			dt.active = false;
			while (!rc_s && dt.next < nb) {
				if (own < kd) {
					rc_s = emit_stripes(e, Pc, s_ana, own, own + 1);
					own++;
				}
				if (!rc_s)
					rc_s = emit_stripes(e, dt.Pc, dt.s_ana, dt.next, dt.next + 1);
				dt.next++;
			}
			if (!rc_s)
				rc_s = emit_tail(e, dt.Pc, dt.s_ana, dt.s_ent, dt.slot, dt.ei);
		}
		if (!rc_s && own < kd)
			rc_s = emit_stripes(e, Pc, s_ana, own, kd);
		if (rc_s)
			return rc_s;
		if (kd < nb) {
			dt.active = true;
			dt.Pc = Pc;
			dt.s_ana = s_ana;
			dt.s_ent = x.s_ent;
			dt.slot = x.slot;
			dt.ei = ei;
			dt.next = kd;
			flusher_arm(e);
		} else if ((rc_s = emit_tail(e, Pc, s_ana, x.s_ent, x.slot, ei))) {
			return rc_s;
		}
		tail_done = true;
	} else if (Launch::inter(Pc, s_ana, Pc.mb_w * Pc.row_num * Pc.n_streams)) {
		return fail(e, -EIO, "k_inter launch failed");
	}
	if (c == C - 1 && !tail_done)
		tm_end(e, f, CEDAR_ENC_STAGE_INTER, s_ana);
	return 0;
}

// ---- entropy coding of cohort c on its own stream, concurrent with deblocking ----
static int enqueue_entropy(FrameCtx &x, Params &Pc, int c, hipStream_t s_entc)
{
	cedar_enc *e = x.e;
	InFlight &f = *x.f;
	const bool pipe = x.pipe;
	const int S = x.S, C = x.C, ei = x.ds * MAX_COHORTS + c;
	HIP_TRY(e, hipStreamWaitEvent(s_entc, e->ev_analysis[ei], 0));
	if (c == 0)
		tm_begin(e, f, CEDAR_ENC_STAGE_ENTROPY, s_entc);
	// Pipelined: as few packets on the entropy stream as possible (see enqueue_deblock): the size
	// words are left zeroed and the done flag raised by the chain's last kernel
	if (!pipe && !x.self_zero)
		HIP_TRY(e, hipMemsetAsync(Pc.stream_size, 0, sizeof(uint32_t) * (Pc.n_streams + 4), s_entc));
	// macroblocks differ in QP: QP_Y as the decoder will hold it, for the mb_qp_delta of the next one
	if (Pc.roi_off && Launch::qp_eff(Pc, s_entc))
		return fail(e, -EIO, "k_qp_eff launch failed");
	// (frames back to back: every cohort's last kernel leaves its size words zeroed, the last one raises the flag)
	Pc.done_flag = pipe || (x.self_reset && c == C - 1) ? f.d_done : x.self_zero ? e->d_abort_all + 8 : nullptr;
	Pc.done_tag = f.done_tag;
	if (Launch::entropy_out(Pc, s_entc, e->dv_out[f.host_out] + (size_t)(S * c / C) * (C > 1 ? e->stream_out_cap : 0),
				e->dv_sizes[x.slot] + (size_t)c * (S + 2)))
		return fail(e, -EIO, "entropy launch failed");
	if (c == C - 1)
		tm_end(e, f, CEDAR_ENC_STAGE_ENTROPY, s_entc);
	if (!pipe && e->ev_entropy[ei])
		HIP_TRY(e, hipEventRecord(e->ev_entropy[ei], s_entc));
	if (!pipe && c == C - 1) {
		for (int c2 = 0; c2 < c; c2++)
			if (e->ev_entropy[x.ds * MAX_COHORTS + c2] && (e->s_ent_c[c2] ? e->s_ent_c[c2] : x.s_ent) != s_entc)
				HIP_TRY(e, hipStreamWaitEvent(s_entc, e->ev_entropy[x.ds * MAX_COHORTS + c2], 0));
		HIP_TRY(e, hipEventRecord(f.done_ent, s_entc));
		if (!x.self_reset && Launch::flag(s_entc, f.d_done, f.done_tag))
			return fail(e, -EIO, "k_flag launch failed");
	}
	return 0;
}

// One frame of every stream of the session: CEDAR_IOCTL_ENCODE up to the trigger (cedar.c:1032-1176).
int submit_frame(cedar_enc *e, const ReplayIn *replay)
{
	if (e->fl_count >= e->depth)
		return fail(e, -EBUSY, "pipeline full: collect first");
	const double t0 = now_ms();
	HIP_TRY(e, hipSetDevice(e->device));
	const int S = e->ext.n_streams;
	if (e->fl_count == 0)
		e->fl_head = 0; // an empty pipeline always lands in host buffer 0 (the one CONFIG returned)
	const int slot = (e->fl_head + e->fl_count) % e->depth;
	InFlight &f = e->fl[slot];
	memset(f.tm.used, 0, sizeof(f.tm.used));
	f.gop_start = replay ? replay->gop_start : e->next_gop_start;
	e->next_gop_start = false;
	if (f.gop_start) // a group of GOP lanes begins: also when a failed step left the group before it a frame short
		e->frame_p_count = 0;
	f.before = GopState{ e->frame_count, e->frame_p_count, e->cur, e->prev_slot, e->prev2_slot, e->idr_count };
	// frames in flight were submitted in order, so counters advance at submit
	const bool idr = e->frame_p_count == 0; // cedar.c:1047-1050
	// Device-side per-frame state rotates with the frame number (never the slot of the frame
	// before, even when the ring of frames in flight restarts at 0)
	const int ds = e->frame_count % e->depth;
	f.ds = ds;
	const uint32_t seq = ++e->submit_seq;
	f.done_tag = seq;
	FrameCtx x{ e, &f, e->P, S, slot, ds, e->prev_slot, e->cohorts, idr, e->row_pipe };
	Params &P = x.P;
	frame_params(e, f, P, replay, idr, ds, seq);
	frame_streams(x);
	const bool pipe = x.pipe;
	const int C = x.C;

	// ---- input: H2D of the pinned buffers unless the caller supplied device frames ----
	tm_begin(e, f, CEDAR_ENC_STAGE_INPUT, x.s_in);
	bool any_host = false; // (something of this frame came from the pinned buffers)
	{
		const int rc_in = submit_input(e, f, P, replay, ds, x.s_in, x.in_work, any_host);
		if (rc_in)
			return rc_in;
	}
	// (k_input -- k_intra_decide without it -- also zeroes the ticket / abort / progress words of the intra wavefront)
	*f.h_wf_err = 0;
	P.wf_err = f.d_wf_err;
	P.hdr = f.dv_hdr; // a few words per frame: the kernels read them from pinned host memory
	P.prefix_len = f.dv_prefix;
	build_frame_headers(e, f, P, idr);
	entropy_lane_params(e, P, x.lane, idr);
	f.lane = x.lane;
	f.host_out = slot; // every frame in flight owns one host buffer until it is collected
	if (e->ext.scene_cut > 0)
		e->h_sizes[slot][S + 2] = 0; // (k_copy_out reports a P picture's votes there)
	f.s_ent = x.s_ent;
	f.s_db = x.s_db;
	f.events = !pipe;
	double t_main = t0, t_ent = t0;

	// The streams of a big batch go through the stages in COHORTS (sessions whose frames run back to
	// back): a P picture's search waits for the deblocking of the picture before it, and with all
	// streams in one launch the GPU does search, THEN deblocking -- the one fills the chip with SAD
	// issue, the other is a dependency chain on a few waves per CU. Two cohorts enqueue
	//   analysis stream   search(A, n)   search(B, n)   [db(A, n)] search(A, n+1)  [db(B, n)] search(B, n+1)
	//   deblock stream    [an(A, n)] db(A, n)           [an(B, n)] db(B, n)
	// so cohort A's deblocking runs beside cohort B's search. A cohort is the same launches on
	// offset pointers; scratch that a launch counts or compacts in (tickets, size words, the packed
	// output) is per cohort.
	// Frames back to back: sync words and size words are left zeroed, the completion flags raised, by the kernels
	// themselves as for pipelined frames (not with cohorts on streams of their own: the last cohort's kernel
	// would speak for kernels of other streams)
	x.self_zero = !pipe && e->batch_self_reset;                      // the kernels leave their sync / size words zeroed
	x.self_reset = x.self_zero && !e->s_db_c[1] && !e->s_ent_c[1];   // ... and raise the completion flags too
	// a held-back tail only waits for the first stripes of a pipelined P picture; anything else: out with it
	if (e->deferred.active && !(pipe && !idr && C == 1)) {
		const int rc_f = flush_deferred(e);
		if (rc_f)
			return rc_f;
	}
	for (int c = 0; c < C; c++) {
		int rc = 0;
		bool tail_done = false;
		Params Pc = cohort_params(e, P, c);
		const int ei = ds * MAX_COHORTS + c;
		// the streams of this cohort's deblocking and entropy chain (cedar_enc::s_db_c)
		hipStream_t s_dbc = !pipe && !x.ahead && e->s_db_c[c] ? e->s_db_c[c] : x.s_db;
		hipStream_t s_entc = !pipe && e->s_ent_c[c] ? e->s_ent_c[c] : x.s_ent;
		if (!Pc.src_direct && Launch::input(Pc, x.s_in))
			return fail(e, -EIO, "k_input launch failed");
		if (c == C - 1)
			tm_end(e, f, CEDAR_ENC_STAGE_INPUT, x.s_in);
		// (frames read where the caller put them and nothing copied: the input stream has nothing of this frame to wait for)
		if (!Pc.src_direct || x.in_work) {
			HIP_TRY(e, hipEventRecord(e->ev_input[ei], x.s_in));
			if (x.s_in != x.s_ana)
				HIP_TRY(e, hipStreamWaitEvent(x.s_ana, e->ev_input[ei], 0));
		}
		// Row-pipelined, the deblock kernel only polls progress words: it is queued BEFORE this
		// frame's analysis so that it is resident and waiting when the first stripe arrives.
		if (pipe && !x.db_late && (rc = enqueue_deblock(x, Pc, c, s_dbc)))
			return rc;
		if ((rc = enqueue_analysis(x, Pc, c, tail_done)))
			return rc;
		if (tail_done) { // a pipelined P picture: emit_tail has (or will have) done the rest
			t_main = t_ent = now_ms();
			continue;
		}
		HIP_TRY(e, hipEventRecord(e->ev_analysis[ei], x.s_ana));
		t_main = now_ms();
		// (frames back to back: CEDAR_DB_FIRST enqueues the deblocking BEFORE the entropy chain)
		const bool db_first = (!pipe || x.db_late) && e->db_first;
		if (db_first && (rc = enqueue_deblock(x, Pc, c, s_dbc)))
			return rc;
		if ((rc = enqueue_entropy(x, Pc, c, s_entc)))
			return rc;
		t_ent = now_ms();
		if ((!pipe || x.db_late) && !db_first && (rc = enqueue_deblock(x, Pc, c, s_dbc)))
			return rc;
	}

	f.active = true;
	f.idr = idr;
	e->fl_count++;
	e->last_slot = ds;
	e->last_rec = e->cur;
	e->last_lane = x.lane;
	e->prev2_slot = e->prev_slot;
	e->prev_slot = ds;
	// cedar.c:1193-1201: advance the GOP counter, swap reference frames
	e->frame_p_count++;
	if (e->frame_p_count == e->cfg.keyframe_interval)
		e->frame_p_count = 0;
	e->frame_count++;
	if (idr)
		e->idr_count++;
	e->cur = (e->cur + 1) % e->nrec;
	if (any_host) // the caller may refill the pinned input buffers once they are on the device
		HIP_TRY(e, hipEventSynchronize(e->ev_host_in));
	if (getenv("CEDAR_TRACE_SUBMIT"))
		fprintf(stderr, "submit frame %d: main %.3f ms, entropy %.3f ms, deblock %.3f ms\n", e->frame_count - 1, t_main - t0, t_ent - t_main, now_ms() - t_ent);
	e->stats.encode_wall_ms += now_ms() - t0;
	return 0;
}

// A frame failed (cedar.c:1182-1191: time-out or bad status -> -1 before the GOP counter and the
// reference swap, :1193-1201). The reference's caller then goes on with the next frame
// (h264enc.c:189-192). With frames in flight "the next frame" has already been submitted on top
// of the broken one: drain the device, put the GOP / reference state back to what it was before
// the failed frame, and encode the frames behind it again from the inputs the session still holds.
static int recover(cedar_enc *e, const InFlight &failed, const ReplayIn *again = nullptr)
{
	flush_deferred(e); // (the device must be able to drain; whatever this enqueues is dropped with the rest)
	hipDeviceSynchronize();
	std::vector<ReplayIn> later;
	if (again) // the failed frame itself goes first (requantised, see collect_frame)
		later.push_back(*again);
	// a plain session's frames behind a dropped one move up by one frame index -- up to the next frame
	// that was given its index explicitly (cedar_enc_set_frame_index: it starts a GOP where it is)
	long shift = (again || e->mux) ? 0 : 1;
	for (int k = 0; k < e->fl_count; k++) {
		InFlight &g = e->fl[(e->fl_head + k) % e->depth];
		if (g.gop_start && !e->mux)
			shift = 0;
		std::vector<long> gidx = g.lane_gidx;
		for (long &v : gidx)
			v = v < 0 ? v : v - shift;
		later.push_back(ReplayIn{ g.in_used, g.in_own, gidx, g.gop_start });
		g.active = false;
	}
	const GopState &b = failed.before;
	e->frame_count = b.frame_count;
	e->frame_p_count = b.frame_p_count;
	e->cur = b.cur;
	// The device has drained: the reconstruction the first replayed frame searches is final, so it waits
	// for no deblock progress at all. (It must not: with the ring full the frame depth - 1 behind the
	// failed one has taken the slot -- and the progress tag -- of the frame before it, and a gate asked to
	// wait for that newer tag on words that still hold the older one would spin until its time-out.)
	e->prev_slot = -1;
	e->prev2_slot = -1;
	e->idr_count = b.idr_count;
	e->fl_count = 0;
	e->fl_head = 0;
	// tickets, abort flags and hand-off progress of the wavefront kernels; the size / error words of
	// the entropy lanes (a kernel that gave up leaves them as they were)
	const int n_units = e->ext.n_streams * e->ext.slice_num;
	(void)n_units;
	for (int i = 0; i < e->depth; i++) {
		HIP_TRY(e, hipMemset(e->d_wf_sync[i], 0, sizeof(uint32_t) * MAX_COHORTS * 3 * WF_SYNC_WORDS));
		*e->fl[i].h_wf_err = 0;
	}
	for (int l = 0; l < e->ent_lanes; l++)
		HIP_TRY(e, hipMemset(e->d_stream_size[l], 0, sizeof(uint32_t) * MAX_COHORTS * (e->ext.n_streams + 4)));
	for (uint32_t *v : e->d_cut_votes) // (the frame that failed never reported, and so never cleared, its votes)
		if (v)
			HIP_TRY(e, hipMemset(v, 0, sizeof(uint32_t)));
	HIP_TRY(e, hipMemset(e->d_abort_all, 0, sizeof(uint32_t)));
	HIP_TRY(e, hipStreamSynchronize(nullptr)); // (hipMemset runs on the null stream, which the session's streams do not wait for)
	for (const ReplayIn &r : later) {
		const int rc = submit_frame(e, &r);
		if (rc)
			return rc;
	}
	return 0;
}

// Development (CEDAR_DEBUG_HANG): what the progress and ticket words, and the device's own event log, say when a hand-off of
// frame f timed out.
// CEDAR_DEBUG_HANG: the device's own account of what its kernels did -- code, frame tag, argument, microseconds since the first
// entry -- printed when a hand-off times out, or when the session closes if CEDAR_DEBUG_LOG_DUMP is set (tools/dev/loop_timeline.py).
void dump_event_log(cedar_enc *e)
{
	if (e->d_dbg_log) { // the device's own account: code, frame tag, argument, microseconds since the first entry
		std::vector<uint32_t> lg((size_t)4 + 4 * DBG_LOG_ENTRIES);
		if (hipMemcpy(lg.data(), e->d_dbg_log, lg.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
			const size_t n = std::min<size_t>(lg[0], DBG_LOG_ENTRIES);
			std::vector<std::array<uint64_t, 3>> ev;
			for (size_t i = 0; i < n; i++)
				ev.push_back({ (uint64_t)lg[4 + 4 * i + 2] | (uint64_t)lg[4 + 4 * i + 3] << 32, lg[4 + 4 * i], lg[4 + 4 * i + 1] });
			std::sort(ev.begin(), ev.end());
			static const char *const names[12] = { "?", "input", "intra_decide", "intra_rows start", "intra_rows end", "deblock start", "deblock end", "gate start",
							       "gate end", "inter", "inter tail wait", "inter tail wait over" };
			fprintf(stderr, " device log, %zu entries (us since the first | event | frame tag | argument):\n", n);
			for (size_t i = 0; i < n; i++) {
				const uint32_t code = (uint32_t)ev[i][1] & 0xff, tag = (uint32_t)ev[i][1] >> 8, arg = (uint32_t)ev[i][2];
				if (code == 3 || code == 4 || code == 5 || code == 6) {
					if ((arg & 0xff) != 0 && (arg & 0xff) != (uint32_t)e->db_bands - 1 && (arg & 0xff) != (uint32_t)e->db_bands / 2)
						continue; // (first, middle and last band of a picture say enough)
				}
				fprintf(stderr, "  %10.1f %-16s tag %3u arg %08x\n", (double)(ev[i][0] - ev[0][0]) / 100.0, names[code < 12 ? code : 0], tag, arg);
			}
		}
	}
}

static void dump_hang_state(cedar_enc *e, const InFlight &f, uint32_t wf_err)
{
	const int S = e->ext.n_streams;
	hipDeviceSynchronize();
	const int n_units = S * e->ext.slice_num;
	std::vector<uint32_t> w((size_t)16 + 4 * n_units * PROG_BANDS), sy((size_t)3 * WF_SYNC_WORDS);
	fprintf(stderr, "HANG: wf_err %u, frame idr %d, seq/tag %u, ds %d, h_done %u %u, frames in flight behind it %d, frame_count %d\n", wf_err, (int)f.idr, f.done_tag, f.ds,
		f.h_done[0], f.h_done[1], e->fl_count, e->frame_count);
	uint32_t ab[8] = {};
	if (hipMemcpy(ab, e->d_abort_all, sizeof(ab), hipMemcpyDeviceToHost) == hipSuccess)
		fprintf(stderr, " first wait that ran out: band %u planes %u idr %u unit %u | need %u saw %08x %08x %08x at step %u, tag %08x | %dx%d mb, db_bands %d\n", (ab[1] - 1) & 0xff, (ab[1] >> 8) & 15,
			(ab[1] >> 12) & 1, ab[1] >> 16, ab[2], ab[3], ab[4], ab[5], ab[6], ab[7], e->P.mb_w, e->P.mb_h, e->db_bands);
	dump_event_log(e);
	for (int k = 0; k < e->depth; k++) {
		if (hipMemcpy(w.data(), e->d_prog[k], w.size() * 4, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(sy.data(), e->d_wf_sync[k], sy.size() * 4, hipMemcpyDeviceToHost) != hipSuccess)
			break;
		fprintf(stderr, " slot %2d%s tag %08x: inter %08x | db_y", k, k == f.ds ? "*" : " ", e->slot_tag[k], w[0]);
		for (int b = 0; b < std::min(e->db_bands, 20); b++) fprintf(stderr, " %x", w[16 + b]);
		fprintf(stderr, " | db_c");
		for (int b = 0; b < std::min(e->db_bands, 20); b++) fprintf(stderr, " %x", w[16 + n_units * PROG_BANDS + b]);
		fprintf(stderr, " | intra_y");
		for (int b = 0; b < std::min(e->db_bands, 20); b++) fprintf(stderr, " %x", w[16 + 2 * n_units * PROG_BANDS + b]);
		fprintf(stderr, " | intra_c");
		for (int b = 0; b < std::min(e->db_bands, 20); b++) fprintf(stderr, " %x", w[16 + 3 * n_units * PROG_BANDS + b]);
		fprintf(stderr, " | sync0 %u %u %u %u sync1 %u %u %u sync2 %u %u %u\n", sy[0], sy[1], sy[2], sy[3], sy[WF_SYNC_WORDS], sy[WF_SYNC_WORDS + 1], sy[WF_SYNC_WORDS + 2],
			sy[2 * WF_SYNC_WORDS], sy[2 * WF_SYNC_WORDS + 1], sy[2 * WF_SYNC_WORDS + 2]);
	}
}

// "wait for the IRQ" (cedar.c:1178-1180). False: the device reported an error.
static bool wait_frame(InFlight &f)
{
	bool hip_failed = false;
	// poll the frame's own completion flags (see InFlight::h_done); fall back to the events
	// if they do not show up (they are written after everything the events cover)
	const double t_poll = now_ms();
	unsigned spins = 0;
	while (f.h_done[0] != f.done_tag || f.h_done[1] != f.done_tag) {
		if ((++spins & 63) == 0) {
			if (now_ms() - t_poll > 2000.0)
				break;
			sched_yield();
		}
	}
	if (f.h_done[0] != f.done_tag || f.h_done[1] != f.done_tag) {
		// (pipelined frames record no events: their streams are kept to the fewest packets)
		hipError_t r1 = f.events ? hipEventSynchronize(f.done_ent) : hipStreamSynchronize(f.s_ent);
		hipError_t r2 = f.events ? hipEventSynchronize(f.done_main) : hipStreamSynchronize(f.s_db);
		hip_failed = r1 != hipSuccess || r2 != hipSuccess;
	}
	std::atomic_thread_fence(std::memory_order_acquire);
	return !hip_failed;
}

// where every stream's access unit lies in the frame's host buffer, with the AUD / SPS / PPS bytes the host writes in front
static void place_outputs(cedar_enc *e, const InFlight &f, int slot)
{
	const int S = e->ext.n_streams, C = e->cohorts;
	for (int c = 0; c < C; c++) {
		const int s0 = S * c / C, s1 = S * (c + 1) / C;
		const uint32_t *sz = e->h_sizes[slot] + (size_t)c * (S + 2);
		uint32_t off = (uint32_t)((size_t)s0 * (C > 1 ? e->stream_out_cap : 0));
		for (int s = s0; s < s1; s++) {
			e->res_off[s] = off;
			e->res_size[s] = sz[s - s0];
			e->pk_seen[f.idr ? 1 : 0] = std::max(e->pk_seen[f.idr ? 1 : 0], (int)(sz[s - s0] / PK_CHUNK) + 1);
			uint8_t *front = e->h_out[f.host_out] + off;
			if (f.aud) {
				const std::vector<uint8_t> &a = e->aud_nal[f.aud - 1];
				memcpy(front, a.data(), a.size());
				front += a.size();
			}
			if (f.with_ps[s])
				memcpy(front, e->sps_pps.data(), e->sps_pps.size());
			off += sz[s - s0];
		}
	}
}

// what the rate law and the stage timers keep of a frame that worked
static void account_frame(cedar_enc *e, InFlight &f, uint32_t total)
{
	const int S = e->ext.n_streams;
	e->rc_bits[f.before.frame_count & 63] = (int64_t)total * 8;
	e->rc_qp[f.before.frame_count & 63] = f.qp;
	if (e->rc_lanes) { // (a lock-step step = one frame of every lane of its group)
		const long m = f.before.frame_count / e->cfg.keyframe_interval;
		if (e->grp_id1[m & 31] != m + 1) {
			e->grp_id1[m & 31] = m + 1;
			e->grp_sum[m & 31] = 0;
			e->grp_n[m & 31] = 0;
		}
		for (int s = 0; s < S; s++)
			if (f.lane_gidx[s] >= 0) { // (idle lanes of the last group code a frame nobody reads)
				e->grp_sum[m & 31] += ((int64_t)e->res_size[s] * 8 * rc_pow2_6_q16(f.qp - e->cfg.qp)) >> 16;
				e->grp_n[m & 31]++;
			}
	}
	if (e->rc_multi)
		for (int s = 0; s < S; s++) {
			e->rc_bits_s[(size_t)s * 64 + (f.before.frame_count & 63)] = (int64_t)e->res_size[s] * 8;
			e->rc_qp_s[(size_t)s * 64 + (f.before.frame_count & 63)] = f.h_qp_s[s];
		}
	e->stats.frame_qp = f.qp;
	if (e->profiling) {
		static const char *const stage_name[CEDAR_ENC_STAGE_COUNT] = { "input", "intra_decide", "intra_recon", "inter", "deblock", "entropy" };
		const bool timeline = e->ev_base && getenv("CEDAR_TRACE_TIMELINE");
		for (int st = 0; st < CEDAR_ENC_STAGE_COUNT; st++)
			if (f.tm.used[st]) {
				float ms = 0;
				// (a pipelined frame's done flags are raised from inside its last kernels: the
				// marker behind such a kernel may be a few microseconds away still)
				hipEventSynchronize(f.tm.ev[st][1]);
				if (hipEventElapsedTime(&ms, f.tm.ev[st][0], f.tm.ev[st][1]) == hipSuccess) {
					e->stats.ms[st] += ms;
					e->stats.launches[st]++;
				}
				float a = 0, b = 0; // when the stage ran on the device, relative to the session's base event
				if (timeline && hipEventElapsedTime(&a, e->ev_base, f.tm.ev[st][0]) == hipSuccess &&
				    hipEventElapsedTime(&b, e->ev_base, f.tm.ev[st][1]) == hipSuccess)
					fprintf(stderr, "timeline frame %llu %s %.3f %.3f\n", (unsigned long long)e->stats.frames, stage_name[st], a, b);
			}
	}
}

int collect_frame(cedar_enc *e)
{
	if (e->fl_count == 0)
		return fail(e, -EINVAL, "nothing in flight");
	const double t0 = now_ms();
	const int S = e->ext.n_streams;
	const int slot = e->fl_head;
	InFlight &f = e->fl[slot];
	HIP_TRY(e, hipSetDevice(e->device));
	if (e->deferred.active && e->deferred.slot == slot) { // the newest frame is the one being waited for
		const int rc_f = flush_deferred(e);
		if (rc_f)
			return rc_f;
	}
	const bool hip_failed = !wait_frame(f);
	e->fl_head = (e->fl_head + 1) % e->depth;
	e->fl_count--;
	f.active = false;
	// sizes per cohort: [streams of the cohort] total, error flags
	const int C = e->cohorts;
	uint32_t total = 0, flags = 0;
	bool too_big = false;
	for (int c = 0; c < C; c++) {
		const int n = S * (c + 1) / C - S * c / C;
		const uint32_t *sz = e->h_sizes[slot] + (size_t)c * (S + 2);
		total += sz[n];
		flags |= sz[n + 1];
		too_big = too_big || sz[n] > (size_t)n * e->stream_out_cap;
	}
	const uint32_t wf_err = *f.h_wf_err;
	if (wf_err && getenv("CEDAR_DEBUG_HANG"))
		dump_hang_state(e, f, wf_err);
	// Scene cut (cedar_enc_config_ext.scene_cut): enough macroblocks of this P picture said their own mean predicts them
	// better than the best vector: the picture is coded again as an IDR picture that starts a new GOP, and the frames
	// submitted behind it after it -- the machinery of a failed frame, with the frame itself put back first.
	if (e->ext.scene_cut > 0 && !f.idr && !hip_failed && !wf_err && !(flags & ~4u) && !too_big &&
	    (uint64_t)e->h_sizes[slot][S + 2] * 100 >= (uint64_t)e->ext.scene_cut * e->P.n_mb) {
		// (no QP bump survives the restart -- the twin's `goto restart` picks the picture's base QP again. Not that a bumped
		// pass can get here: a vote needs SADmean < min over vectors (SAD + lambda (bits - 32)) with bits <= 30, which only
		// falls as lambda grows, so a picture that did not cut at its first QP does not cut 6 steps coarser.)
		ReplayIn again{ f.in_used, f.in_own, f.lane_gidx, true, 0 };
		const InFlight snapshot = f;
		const int rc = recover(e, snapshot, &again);
		if (rc)
			return rc;
		e->stats.scene_cuts++;
		return collect_frame(e);
	}
	// flag 4: a macroblock of more than 3200 bits (H.264 A.3.1). Counted, or -- strict_mb_bits -- refused
	const bool oversize = (flags & 4u) != 0;
	flags &= ~4u;
	if (oversize && !e->ext.strict_mb_bits)
		e->stats.oversize_frames++;
	if (oversize && e->ext.strict_mb_bits == 2 && !hip_failed && !wf_err && !flags && !too_big && f.qp + 6 <= 51) {
		// The hard cap: the picture is coded again 6 QP steps coarser (slice_qp_delta), as often as it
		// takes -- the frames submitted behind it run again on the new reconstruction. Noise at a very
		// low QP is what gets here; everything else never does.
		ReplayIn again{ f.in_used, f.in_own, f.lane_gidx, f.gop_start, f.qp_bump + 6 };
		const InFlight snapshot = f;
		const int rc = recover(e, snapshot, &again);
		if (rc)
			return rc;
		e->stats.requantised_frames++;
		return collect_frame(e);
	}
	if (hip_failed || wf_err || flags || too_big || (oversize && e->ext.strict_mb_bits)) {
		// this frame is dropped, the session is where it was before it, the frames behind it run again
		const int rc = recover(e, f);
		if (rc)
			return rc;
		if (hip_failed)
			return fail(e, -EIO, "the device reported an error for this frame");
		if (wf_err)
			return fail(e, -EIO, "wavefront hand-off timed out (flags %u)", wf_err);
		if (oversize && !flags && !too_big)
			return fail(e, -EOVERFLOW, "a macroblock exceeds 3200 bits (H.264 A.3.1): raise the QP");
		return fail(e, -ENOSPC, "bytestream buffer too small (flags %u, %u bytes)", flags, total);
	}
	place_outputs(e, f, slot);
	e->res_host = f.host_out;
	e->last_total = total;
	account_frame(e, f, total);
	e->stats.frames++;
	e->stats.encode_wall_ms += now_ms() - t0;
	return (int)total; // cedar.c:1207-1208: bytes now valid at bytestream offset 0
}

// ---- gop_parallel: S streams x G closed GOPs side by side (see GopMux) ----

} // namespace cedar
