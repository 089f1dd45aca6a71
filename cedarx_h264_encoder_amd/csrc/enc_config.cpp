// cedarx_h264_encoder_amd/csrc/enc_config.cpp
// CEDAR_IOCTL_CONFIG (cedar.c:732-866) and the buffer manager behind it (cedar.c:605-704): parameter checks, host and
// device buffers, streams and the roles they get, the schedule the policy of enc_plan.h picks, and the one pipelined
// session per device across sessions and processes.
#include "enc_session.h"

namespace cedar {

// One pipelined session per device and process: two of them would queue polling kernels of
// one behind polling kernels of the other on shared hardware queues and wait for each other
// until the spin bounds. Further sessions run their frames back to back.
std::atomic<int> g_pipelined_sessions[64];
// ... and per device across PROCESSES: a pipelined session's deblock workgroups sit on their CUs and poll until their
// picture's analysis comes (enc_plan.h counts them per XCD against the CUs of the device), and that count knows nothing
// of another process whose resident workgroups take the same CUs. The session that pipelines holds an exclusive
// advisory lock on a file named after the device's PCI bus id for as long as it lives; a session that finds it taken
// runs its frames back to back (nothing resident waits for anything there), exactly as the second session of one
// process does. Where no lock file can be made the per-process rule is all there is.
int pipe_lock_acquire(int device)
{
	char bus[64] = "unknown";
	if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess)
		snprintf(bus, sizeof(bus), "dev%d", device);
	for (char *c = bus; *c; c++)
		if (*c == ':' || *c == '/')
			*c = '_';
	const char *dir = getenv("CEDAR_LOCK_DIR");
	char path[256];
	snprintf(path, sizeof(path), "%s/cedar_h264enc.%s.lock", dir && *dir ? dir : "/tmp", bus);
	const int fd = open(path, O_RDWR | O_CREAT | O_CLOEXEC, 0666);
	if (fd < 0)
		return -2; // no lock file: the per-process rule only
	if (flock(fd, LOCK_EX | LOCK_NB) != 0) {
		close(fd);
		return -1; // another process pipelines on this device
	}
	return fd;
}
void pipe_token_release(int device, bool &holds, int &lock_fd)
{
	if (holds)
		g_pipelined_sessions[device & 63].fetch_sub(1);
	holds = false;
	if (lock_fd >= 0)
		close(lock_fd); // (drops the lock)
	lock_fd = -1;
}

void free_all(cedar_enc *e)
{
	flusher_stop(e); // (before anything the thread could touch goes away)
	auto F = [](void *p) { if (p) hipFree(p); };
	auto H = [](void *p) { if (p) hipHostFree(p); };
	F(e->d_in_y); F(e->d_in_uv); F((void *)e->d_in_ptrs);
	F(e->d_roi_off); F(e->d_qtab);
	for (int i = 0; i < CEDAR_ENC_PIPELINE_DEPTH + 1; i++) {
		F(e->d_rec_y[i]); F(e->d_rec_uv[i]);
	}
	for (int i = 0; i < CEDAR_ENC_PIPELINE_DEPTH; i++) {
		F(e->d_src_y[i]); F(e->d_src_uv[i]); F(e->d_mbi[i]); F(e->d_i4[i]); F(e->d_nnz[i]); F(e->d_coef[i]);
		F(e->d_qp_eff[i]); F(e->d_qp_t[i]);
		F(e->d_wf_sync[i]); F(e->d_wf_hbuf[i]); F(e->d_db_hbuf[i]); F(e->d_ir_hbuf[i]); F(e->d_prog[i]); F(e->d_out[i]);
	}
	for (int l = 0; l < ENT_LANES; l++) {
		F(e->d_slot[l]); F(e->d_mb_bits[l]); F(e->d_skip[l]); F(e->d_bit_off[l]); F(e->d_run[l]); F(e->d_unit_bits[l]);
		F(e->d_ent_rec[l]); F(e->d_pk_agg[l]); F(e->d_pk_unit[l]); F(e->d_pk_first[l]);
		F(e->d_rbsp[l]); F(e->d_esc[l]); F(e->d_nal_off[l]); F(e->d_nal_size[l]); F(e->d_stream_size[l]);
	}
	F(e->d_pred4); F(e->d_dummy_rec); F(e->d_stripe_tab); F(e->d_abort_all); F(e->d_dbg_log);
	for (uint32_t *v : e->d_cut_votes)
		F(v);
	if (e->mux) {
		F(e->mux->d_store_y); F(e->mux->d_store_uv);
		delete e->mux;
		e->mux = nullptr;
	}
	H(e->h_luma); H(e->h_chroma);
	for (int i = 0; i < CEDAR_ENC_PIPELINE_DEPTH; i++) {
		H(e->h_out[i]); H(e->h_sizes[i]); H(e->fl[i].h_hdr); H(e->fl[i].h_prefix); H(e->fl[i].h_qp_s); H(e->fl[i].h_wf_err); H((void *)e->fl[i].h_done);
		if (e->fl[i].done_main)
			hipEventDestroy(e->fl[i].done_main);
		if (e->fl[i].done_ent)
			hipEventDestroy(e->fl[i].done_ent);
		for (int c = 0; c < MAX_COHORTS; c++) {
			const int k = i * MAX_COHORTS + c;
			if (e->ev_input[k]) hipEventDestroy(e->ev_input[k]);
			if (e->ev_analysis[k]) hipEventDestroy(e->ev_analysis[k]);
			if (e->ev_deblocked[k]) hipEventDestroy(e->ev_deblocked[k]);
		}
		for (int s = 0; s < CEDAR_ENC_STAGE_COUNT; s++)
			for (int k = 0; k < 2; k++)
				if (e->fl[i].tm.ev[s][k])
					hipEventDestroy(e->fl[i].tm.ev[s][k]);
	}
	if (e->ev_host_in) hipEventDestroy(e->ev_host_in);
	if (e->ev_base) hipEventDestroy(e->ev_base);
	if (e->s_main) hipStreamDestroy(e->s_main);
	for (int i = 1; i < AN_STREAMS; i++)
		if (e->s_ana[i]) hipStreamDestroy(e->s_ana[i]);
	if (e->s_ent) hipStreamDestroy(e->s_ent);
	for (int i = 0; i < DB_STREAMS; i++)
		if (e->s_db[i]) hipStreamDestroy(e->s_db[i]);
	for (int c = 1; c < MAX_COHORTS; c++) // (s_db_c are aliases of s_db[])
		if (e->s_ent_c[c]) hipStreamDestroy(e->s_ent_c[c]);
	for (hipEvent_t ev : e->ev_entropy)
		if (ev) hipEventDestroy(ev);
}

// Which stream does what. Hardware queues share command-processor pipes in pairs, and a queue
// that holds a running kernel with more packets behind it ("blocked") makes its pipe serve the
// other queue slowly: ~21 us per launch instead of ~3 us when the blocked queue is the later of
// the pair, ~6 us when it is the earlier (measured; Launch::probe_pipe, tools/dev/pipe_probe.hip).
// Roles differ in how often they are blocked and in how many dependent launches per frame they
// have on the critical path:
//   MAIN  gates + analysis: a dozen launches per frame, the gates spin
//   ENT   ten short kernels per frame behind an event wait
//   DB0/1 deblock kernels, resident for a frame time; memset + kernel + flag every other frame
//   IDR   an IDR picture's milliseconds of intra reconstruction (look-ahead), once per GOP
//   IN    k_input: never waits, nothing waits for it in steady state
// All ordered pairs of the seven streams are measured and the roles dealt to minimise the
// expected launch delay per frame; the spare stream keeps a pipe-mate company.
int assign_stream_roles(cedar_enc *e, int db_streams_wanted)
{
	enum { MAIN, ENT, DB0, DB1, IDR, IN, DB2, NROLE };
	const bool three = db_streams_wanted >= 3; // a third deblock stream, or a spare
	// (the seventh role is a third deblock stream, a second entropy lane, or -- the default -- the
	// stream that carries every other IDR picture)
	const bool ent1 = !three && e->ent_lanes > 1, idr2 = !three && !ent1;
	const double blocked_frac[NROLE] = { 0.5, 0.3, 0.9, 0.9, idr2 ? 0.25 : 0.45, 0.0, three ? 0.9 : ent1 ? 0.3 : 0.25 };
	const double launches[NROLE] = { 12, ent1 ? 5.0 : 10.0, three ? 1.0 : 1.5, three ? 1.0 : 1.5, idr2 ? 0.15 : 0.3, 0.1,
					 three ? 1.0 : ent1 ? 5.0 : 0.15 };
	hipStream_t pool[7] = { e->s_main, e->s_ent, e->s_db[0], e->s_db[1], e->s_db[2], e->s_ana[1], e->s_ana[2] };
	const int NP = 7;
	double slow[NP][NP] = {}; // [blocked][chain]: extra us per launch
	double base = 0, t = 0;
	for (int k = 0; k < 2; k++)
		if (Launch::probe_pipe(nullptr, pool[0], e->d_prog[0], 24, &base))
			return fail(e, -EIO, "pipe probe failed");
	const bool trace = getenv("CEDAR_TRACE_SUBMIT") != nullptr;
	for (int i = 0; i < NP; i++)
		for (int j = 0; j < NP; j++) {
			if (i == j)
				continue;
			double worst = 0; // the effect comes and goes from one try to the next: keep the worst of three
			for (int k = 0; k < 3; k++) {
				if (Launch::probe_pipe(pool[i], pool[j], e->d_prog[0], 24, &t))
					return fail(e, -EIO, "pipe probe failed");
				worst = t > worst ? t : worst;
			}
			slow[i][j] = worst > base + 1.5 ? worst - base : 0.0;
		}
	// Two queues share a pipe or they do not: what one direction shows holds for the other. A single probe misses
	// the effect now and then (it comes and goes), and a pair measured as free in one direction once dealt the
	// analysis stream a pipe-mate -- with 7 streams on 8 queues exactly one stream has none, and that one must
	// carry the gates: one 1080p stream ran at 27-28 M MB/s instead of 43 M in 2 of 14 sessions.
	for (int i = 0; i < NP; i++)
		for (int j = i + 1; j < NP; j++)
			slow[i][j] = slow[j][i] = slow[i][j] > slow[j][i] ? slow[i][j] : slow[j][i];
	HIP_TRY(e, hipMemset(e->d_prog[0], 0, 64));
	HIP_TRY(e, hipStreamSynchronize(nullptr)); // (the null stream's clear before the probes' own streams touch the words)
	int best[NROLE] = { 0, 1, 2, 3, 5, 6, 4 }, cur[NROLE];
	double best_cost = 1e30;
	bool taken[NP] = {};
	// (CEDAR_STREAM_PRIO=1, setup_streams: roles in classes, the analysis on pool[0])
	static const int cls_of_stream[7] = { 1, 1, -1, -1, -1, 0, 0 };
	static const int cls_of_role[NROLE] = { 1, 1, -1, -1, 0, 0, -1 }; // MAIN ENT DB0 DB1 IDR IN DB2
	struct Rec { // depth-first over the 7! assignments
		const double *blocked_frac, *launches;
		bool classes;
		void go(int r, int *cur, bool *taken, double cost, const double (*slow)[7], int *best, double &best_cost) const
		{
			if (cost >= best_cost)
				return;
			if (r == NROLE) {
				best_cost = cost;
				for (int k = 0; k < NROLE; k++)
					best[k] = cur[k];
				return;
			}
			for (int sidx = 0; sidx < 7; sidx++) {
				if (taken[sidx] || (classes && (cls_of_role[r] != cls_of_stream[sidx] || (r == 0) != (sidx == 0))))
					continue;
				double add = 0;
				for (int q = 0; q < r; q++)
					add += blocked_frac[q] * launches[r] * slow[cur[q]][sidx] + blocked_frac[r] * launches[q] * slow[sidx][cur[q]];
				taken[sidx] = true;
				cur[r] = sidx;
				go(r + 1, cur, taken, cost + add, slow, best, best_cost);
				taken[sidx] = false;
			}
		}
	};
	Rec{ blocked_frac, launches, e->main_prio }.go(0, cur, taken, 0.0, slow, best, best_cost);
	const int spare = best[DB2];
	e->s_main = pool[best[MAIN]];
	e->s_ana[0] = e->s_main;
	e->s_ent = pool[best[ENT]];
	e->s_db[0] = pool[best[DB0]];
	e->s_db[1] = pool[best[DB1]];
	e->s_db[2] = pool[spare];
	e->s_ana[1] = pool[best[IDR]];
	e->s_ana[2] = pool[best[IN]];
	if (trace) {
		fprintf(stderr, "config: launch chain %.1f us per kernel; extra us per launch (blocked -> chain):", base);
		for (int i = 0; i < NP; i++)
			for (int j = 0; j < NP; j++)
				if (slow[i][j] > 0)
					fprintf(stderr, " %d->%d %.1f", i, j, slow[i][j]);
		fprintf(stderr, "; roles main %d ent %d db %d %d idr %d in %d spare %d, expected delay %.1f us per frame\n", best[MAIN],
			best[ENT], best[DB0], best[DB1], best[IDR], best[IN], spare, best_cost);
	}
	return 0;
}


void fill_static_params(cedar_enc *e)
{
	Params &P = e->P;
	const struct cedar_enc_config &c = e->cfg;
	P.width = c.src_width;
	P.height = c.src_height;
	P.cw = c.dst_width;
	P.ch = c.dst_height;
	P.mb_w = P.cw / 16;
	P.mb_h = P.ch / 16;
	P.n_mb = P.mb_w * P.mb_h;
	P.n_streams = e->ext.n_streams;
	P.dummy_mb = (int64_t)e->ext.n_streams * P.n_mb; // one record past the last stream (set again per cohort)
	P.slice_count = e->ext.slice_count;
	P.slice_first = e->ext.slice_first;
	P.slice_num = e->ext.slice_num;
	int r0, r1, q0, q1;
	slice_rows(P.mb_h, P.slice_count, P.slice_first, &r0, &r1);
	slice_rows(P.mb_h, P.slice_count, P.slice_first + P.slice_num - 1, &q0, &q1);
	P.row_first = r0;
	P.row_num = q1 - r0;
	P.search_range = e->ext.search_range;
	P.subpel = e->ext.subpel;
	P.intra_refresh = e->ext.intra_refresh;
	set_qp_params(P, c.qp, e->chroma_qp_offset);
	P.pcm = e->pcm_session && picture_takes_pcm(P.qp, e->chroma_qp_offset) ? 1 : 0;
}

void build_pred4_table(uint8_t tab[9 * 16])
{
	// E[0]=E[1]=l3 E[2]=l2 E[3]=l1 E[4]=l0 E[5]=tl E[6..13]=t0..t7 E[14]=t7
	// kinds: 0 copy E[k]; 1 avg2(E[k],E[k+1]); 2 filt3 centred on E[k]; 3 DC   (clause 8.3.1.2)
	for (int y = 0; y < 4; y++)
		for (int x = 0; x < 4; x++) {
			const int i = 4 * y + x;
			auto set = [&](int mode, int kind, int k) { tab[mode * 16 + i] = (uint8_t)((kind << 4) | k); };
			set(0, 0, 6 + x);
			set(1, 0, 4 - y);
			set(2, 3, 0);
			set(3, 2, 7 + x + y);
			set(4, 2, 5 + x - y);
			{ // vertical-right
				int z = 2 * x - y;
				if (z >= 0 && !(z & 1)) set(5, 1, 5 + x - (y >> 1));
				else if (z >= 0) set(5, 2, 5 + x - (y >> 1));
				else if (z == -1) set(5, 2, 5);
				else set(5, 2, 6 - y);
			}
			{ // horizontal-down
				int z = 2 * y - x;
				if (z >= 0 && !(z & 1)) set(6, 1, 4 - y + (x >> 1));
				else if (z >= 0) set(6, 2, 5 - y + (x >> 1));
				else if (z == -1) set(6, 2, 5);
				else set(6, 2, 4 + x);
			}
			if (!(y & 1)) set(7, 1, 6 + x + (y >> 1));
			else set(7, 2, 7 + x + (y >> 1));
			{ // horizontal-up
				int z = x + 2 * y, k = y + (x >> 1);
				if (z > 5) set(8, 0, 1);
				else if (z == 5) set(8, 2, 1);
				else if (!(z & 1)) set(8, 1, 3 - k);
				else set(8, 2, 3 - k);
			}
		}
}

// geometry + the CEDAR_* development switches of the process -> what the schedule policy (enc_plan.h) works from
static PlanIn plan_input(int mb_w, int mb_h, int slice_count, int slice_first, int slice_num, int n_streams, int cus)
{
	auto env = [](const char *name) { const char *v = getenv(name); return v ? atoi(v) : PLAN_UNSET; };
	const long long slot = (long long)mb_w * mb_h * 1024 + 4096;
	PlanIn in{ mb_w, mb_h, slice_count, slice_first, slice_num, n_streams, cus, (int)(slot < (1ll << 20) ? (1ll << 20) : slot),
		   env("CEDAR_ROW_PIPE"), env("CEDAR_PIPE_DEPTH"), env("CEDAR_WF_SPLIT"), env("CEDAR_WF_SPLIT_INTRA"), env("CEDAR_DB_PAIRS"),
		   env("CEDAR_DEBLOCK_STREAMS"), env("CEDAR_DEBLOCK_PLANES"), env("CEDAR_INTER_BANDS"), env("CEDAR_DEFER"), env("CEDAR_COHORTS"),
		   env("CEDAR_DEBLOCK_ROWS") };
	in.env_intra_rows = env("CEDAR_INTRA_ROWS");
	return in;
}


// CEDAR_IOCTL_CONFIG's parameter checks (cedar.c:749-789) and those of the extension block: pure -- no device, no state but
// the error text. ext has its defaults filled in; S_user / G = the caller's streams and GOP lanes.
static int validate_config(cedar_enc *e, const struct cedar_enc_config *cfg, const struct cedar_enc_config_ext &ext, int S_user, int G)
{
	if ((cfg->src_width & 1) || (cfg->src_height & 1) || cfg->src_width <= 0 || cfg->src_height <= 0)
		return fail(e, -EINVAL, "src width %d, height %d not aligned", cfg->src_width, cfg->src_height);
	if ((cfg->dst_width & 15) || (cfg->dst_height & 15))
		return fail(e, -EINVAL, "dst width %d, height %d not aligned", cfg->dst_width, cfg->dst_height);
	if (cfg->src_width > cfg->dst_width || cfg->src_height > cfg->dst_height)
		return fail(e, -EINVAL, "src (%d,%d) > dst (%d,%d)", cfg->src_width, cfg->src_height, cfg->dst_width, cfg->dst_height);
	if (cfg->dst_width != align_up(cfg->src_width, 16) || cfg->dst_height != align_up(cfg->src_height, 16))
		return fail(e, -EINVAL, "dst must be src rounded up to 16 (no scaler on this path)");
	if (cfg->qp <= 0 || cfg->qp > 47)
		return fail(e, -EINVAL, "invalid QP %d", cfg->qp);
	if (cfg->src_format != CEDAR_ENC_FORMAT_NV12)
		return fail(e, -EINVAL, "invalid color format (nv12 only)");
	if (cfg->keyframe_interval <= 0 || cfg->keyframe_interval >= 32)
		return fail(e, -EINVAL, "invalid keyframe interval %d", cfg->keyframe_interval);
	if (cfg->entropy_coding_mode != CEDAR_ENC_ENTROPY_CAVLC)
		return fail(e, -EINVAL, "CABAC not supported: baseline/CAVLC only");
	const int mb_h = cfg->dst_height / 16, mb_w = cfg->dst_width / 16;
	if (ext.n_streams < 1 || ext.slice_count < 1 || ext.slice_count > mb_h || ext.slice_first < 0 ||
	    ext.slice_num < 1 || ext.slice_first + ext.slice_num > ext.slice_count ||
	    ext.n_streams * ext.slice_num > MAX_UNITS)
		return fail(e, -EINVAL, "bad stream/slice layout");
	if (ext.search_range < 0 || ext.search_range > 16)
		return fail(e, -EINVAL, "search range %d outside 0..16", ext.search_range);
	if (ext.fps_num < 0 || ext.fps_den < 0 || (ext.fps_num > 0) != (ext.fps_den > 0) || ext.fps_num > 0x3fffffff)
		return fail(e, -EINVAL, "frame rate %d/%d: both or neither, time_scale must fit 32 bits", ext.fps_num, ext.fps_den);
	if (e->chroma_qp_offset < -12 || e->chroma_qp_offset > 12) // H.264 7.4.2.2
		return fail(e, -EINVAL, "chroma_qp_index_offset %d outside -12..12", e->chroma_qp_offset);
	if (ext.aud != 0 && ext.aud != 1)
		return fail(e, -EINVAL, "aud must be 0 or 1");
	if (ext.gop_low_latency != 0 && ext.gop_low_latency != 1)
		return fail(e, -EINVAL, "gop_low_latency must be 0 or 1");
	if (ext.subpel < 0 || ext.subpel > 2)
		return fail(e, -EINVAL, "subpel %d outside 0..2", ext.subpel);
	if (ext.intra_refresh < 0 || (ext.intra_refresh > 0 && ext.intra_refresh < 3) || ext.intra_refresh > 65535)
		return fail(e, -EINVAL, "intra_refresh %d: 0 (off) or 3..65535", ext.intra_refresh);
	if (ext.roi_count < 0 || ext.roi_count > 4)
		return fail(e, -EINVAL, "roi_count %d outside 0..4", ext.roi_count);
	for (int k = 0; k < ext.roi_count; k++)
		if (ext.roi[k][0] < 0 || ext.roi[k][1] < 0 || ext.roi[k][2] <= 0 || ext.roi[k][3] <= 0 || ext.roi[k][4] < -24 || ext.roi[k][4] > 24)
			return fail(e, -EINVAL, "region of interest %d: x, y >= 0, width, height > 0 (macroblocks), QP offset -24..24", k);
	for (int k = 0; k < 5; k++)
		if (ext.mad_dqp[k] < -12 || ext.mad_dqp[k] > 12)
			return fail(e, -EINVAL, "mad_dqp[%d] = %d outside -12..12", k, ext.mad_dqp[k]);
	for (int k = 0; k < 4; k++)
		if (ext.mad_th[k] < 0 || ext.mad_th[k] > 255 || (k && ext.mad_th[k] < ext.mad_th[k - 1]))
			return fail(e, -EINVAL, "mad_th: four ascending thresholds 0..255");
	if (ext.strict_mb_bits < 0 || ext.strict_mb_bits > 2)
		return fail(e, -EINVAL, "strict_mb_bits must be 0, 1 or 2");
	if (ext.scene_cut < 0 || ext.scene_cut > 100)
		return fail(e, -EINVAL, "scene_cut is a percentage: 0 (off) .. 100");
	if (ext.scene_cut > 0 && (S_user != 1 || G != 1 || ext.slice_num != ext.slice_count || ext.first_frame_index != 0))
		return fail(e, -EINVAL, "scene_cut: one stream, whole pictures, no gop_parallel, a stream that starts at frame 0");
	if (ext.pcm < -1 || ext.pcm > 0)
		return fail(e, -EINVAL, "pcm must be 0 (I_PCM where a macroblock could pass 3200 bits) or -1 (never)");
	if (ext.strict_mb_bits == 2 && (S_user != 1 || G != 1 || ext.slice_num != ext.slice_count))
		return fail(e, -EINVAL, "requantising oversize pictures wants one stream, whole pictures, no GOP lanes (one QP per launch)");
	if (ext.bitrate < 0 || (ext.bitrate > 0 && (ext.slice_num != ext.slice_count || (G != 1 && S_user != 1))))
		return fail(e, -EINVAL, "rate control wants whole pictures; with GOP lanes (one QP per group of lanes) one stream");
	if (mb_h > 512 * ext.slice_count)
		return fail(e, -EINVAL, "picture too tall for one slice");
	if (ext.first_frame_index < 0 || ext.first_frame_index % cfg->keyframe_interval)
		return fail(e, -EINVAL, "first_frame_index %d is not the start of a GOP (keyframe interval %d)", ext.first_frame_index,
			    cfg->keyframe_interval);
	(void)mb_w;
	return 0;
}

// The pinned host side of the buffer manager (cedar.c:605-704): the caller's input buffers, per ring slot the bytestream and
// size buffers k_pack_out writes, and the few words per frame that kernels read from or write to host memory.
static int alloc_host_buffers(cedar_enc *e, int luma_sz, int chroma_sz, int S_user, int S, int n_units)
{
	HIP_TRY(e, hipHostMalloc((void **)&e->h_luma, (size_t)luma_sz * S_user, hipHostMallocDefault));
	HIP_TRY(e, hipHostMalloc((void **)&e->h_chroma, (size_t)chroma_sz * S_user, hipHostMallocDefault));
	for (int i = 0; i < e->depth; i++) {
		const unsigned out_flags = hipHostMallocMapped;
		HIP_TRY(e, hipHostMalloc((void **)&e->h_out[i], e->out_cap + 16, out_flags));
		HIP_TRY(e, hipHostGetDevicePointer((void **)&e->dv_out[i], e->h_out[i], 0));
		HIP_TRY(e, hipHostMalloc((void **)&e->h_sizes[i], (size_t)MAX_COHORTS * (S + 2) * sizeof(uint32_t), out_flags));
		HIP_TRY(e, hipHostGetDevicePointer((void **)&e->dv_sizes[i], e->h_sizes[i], 0));
	}
	for (int i = 0; i < e->depth; i++) {
		HIP_TRY(e, hipHostMalloc((void **)&e->fl[i].h_hdr, sizeof(SliceHdr) * n_units, hipHostMallocMapped));
		HIP_TRY(e, hipHostGetDevicePointer((void **)&e->fl[i].dv_hdr, e->fl[i].h_hdr, 0));
		HIP_TRY(e, hipHostMalloc((void **)&e->fl[i].h_prefix, sizeof(uint32_t) * S, hipHostMallocMapped));
		HIP_TRY(e, hipHostGetDevicePointer((void **)&e->fl[i].dv_prefix, e->fl[i].h_prefix, 0));
		if (e->rc_multi) {
			HIP_TRY(e, hipHostMalloc((void **)&e->fl[i].h_qp_s, (size_t)S, hipHostMallocMapped));
			HIP_TRY(e, hipHostGetDevicePointer((void **)&e->fl[i].dv_qp_s, e->fl[i].h_qp_s, 0));
		}
		HIP_TRY(e, hipHostMalloc((void **)&e->fl[i].h_wf_err, sizeof(uint32_t), hipHostMallocMapped));
		*e->fl[i].h_wf_err = 0;
		HIP_TRY(e, hipHostGetDevicePointer((void **)&e->fl[i].d_wf_err, e->fl[i].h_wf_err, 0));
		HIP_TRY(e, hipHostMalloc((void **)&e->fl[i].h_done, 2 * sizeof(uint32_t), hipHostMallocMapped));
		e->fl[i].h_done[0] = e->fl[i].h_done[1] = 0;
		HIP_TRY(e, hipHostGetDevicePointer((void **)&e->fl[i].d_done, (void *)e->fl[i].h_done, 0));
	}
	return 0;
}

// Macroblocks that differ in QP (regions of interest, MAD classes, rate control over several streams): the offset of every
// macroblock and one row of derived parameters per QP_Y.
static int alloc_qp_tables(cedar_enc *e, bool row_rule)
{
	Params &P = e->P;
	const struct cedar_enc_config_ext &ext = e->ext;
	int rc = 0;
	// the offset of every macroblock (the first region that covers it; all zero when only the MAD
	// classes move QPs), and one row of derived parameters per QP_Y
	std::vector<int8_t> off((size_t)P.n_mb, 0);
	for (int my = 0; my < P.mb_h; my++)
		for (int mx = 0; mx < P.mb_w; mx++)
			for (int k = 0; k < ext.roi_count; k++) {
				const int *r = ext.roi[k];
				if (mx >= r[0] && mx < r[0] + r[2] && my >= r[1] && my < r[1] + r[3]) {
					off[(size_t)my * P.mb_w + mx] = (int8_t)r[4];
					break;
				}
			}
	std::vector<QpRow> rows(52);
	for (int q = 0; q < 52; q++) {
		QpRow &r = rows[q];
		const int qi = q + e->chroma_qp_offset;
		for (int k = 0; k < 3; k++) {
			r.mf[k] = kQuantMF[q % 6][k];
			r.dv[k] = kDequantV[q % 6][k];
			r.tc0[k] = kTc0[q][k];
		}
		r.qbits = 15 + q / 6;
		r.qshift = q / 6;
		r.lambda = lambda_of_qp(q);
		r.qpc = kChromaQp[qi < 0 ? 0 : qi > 51 ? 51 : qi];
		r.alpha = kAlpha[q];
		r.beta = kBeta[q];
		r.pad = 0;
	}
	if ((rc = dalloc(e, &e->d_roi_off, (size_t)P.n_mb))) return rc;
	if ((rc = dalloc(e, &e->d_qtab, (size_t)52))) return rc;
	HIP_TRY(e, hipMemcpy(e->d_roi_off, off.data(), off.size(), hipMemcpyHostToDevice));
	HIP_TRY(e, hipMemcpy(e->d_qtab, rows.data(), rows.size() * sizeof(QpRow), hipMemcpyHostToDevice));
	P.roi_off = e->d_roi_off;
	P.qtab = e->d_qtab;
	P.row_rule = row_rule ? 1 : 0;
	for (int k = 0; k < 5; k++) {
		P.mad_dqp[k] = ext.mad_dqp[k];
		P.mad_on |= ext.mad_dqp[k] != 0;
		if (k < 4)
			P.mad_th[k] = ext.mad_th[k];
	}
	return 0;
}

// Per entropy lane the bit slots of the CAVLC kernels and what k_scan / k_pack_out (or the round-3 chain) meet in.
static int alloc_entropy_scratch(cedar_enc *e)
{
	Params &P = e->P;
	const struct cedar_enc_config_ext &ext = e->ext;
	const int S = ext.n_streams, n_units = S * ext.slice_num, mb_w = P.mb_w;
	int rc = 0;
	{
		// A second entropy lane (frames alternating between two streams with scratch of their own)
		// is there for the asking; measured, one lane keeps up (1080p: 32.5 M MB/s either way)
		const char *envl2 = getenv("CEDAR_ENTROPY_LANES");
		e->ent_lanes = envl2 && atoi(envl2) > 1 && e->depth > 4 ? ENT_LANES : 1;
	}
	for (int l = 0; l < e->ent_lanes; l++) {
		if ((rc = dalloc(e, &e->d_slot[l], (size_t)P.n_mb * S * SLOT_WORDS))) return rc;
		if ((rc = dalloc(e, &e->d_mb_bits[l], (size_t)P.n_mb * S))) return rc;
		if ((rc = dalloc(e, &e->d_skip[l], (size_t)P.n_mb * S))) return rc;
	}
	int max_rows = 0;
	for (int s = ext.slice_first; s < ext.slice_first + ext.slice_num; s++) {
		int a, b;
		slice_rows(P.mb_h, P.slice_count, s, &a, &b);
		if (b - a > max_rows) max_rows = b - a;
	}
	P.unit_mb_cap = max_rows * mb_w;
	P.rbsp_cap = align_up(P.unit_mb_cap * 1024 + 4096, 256);
	P.esc_chunks_cap = P.rbsp_cap / ESC_CHUNK;
	P.out_cap = (int32_t)e->out_cap;
	e->ent_fused = !(getenv("CEDAR_ENTROPY_FUSED") && atoi(getenv("CEDAR_ENTROPY_FUSED")) == 0);
	e->db_first = getenv("CEDAR_DB_FIRST") && atoi(getenv("CEDAR_DB_FIRST")) != 0;
	e->src_direct_ok = (P.width & 15) == 0 && !(getenv("CEDAR_SRC_DIRECT") && atoi(getenv("CEDAR_SRC_DIRECT")) == 0);
	P.pk_chunks_cap = P.rbsp_cap / PK_CHUNK + 1;
	for (int l = 0; l < e->ent_lanes; l++) {
		if ((rc = dalloc(e, &e->d_bit_off[l], (size_t)n_units * (P.unit_mb_cap + 2)))) return rc;
		if ((rc = dalloc(e, &e->d_run[l], (size_t)n_units * (P.unit_mb_cap + 2)))) return rc;
		if ((rc = dalloc(e, &e->d_unit_bits[l], (size_t)n_units))) return rc;
		if (e->ent_fused) { // k_pack_out: no RBSP buffer; its tagged words start zeroed (dalloc) and never hold another frame's tag
			if ((rc = dalloc(e, &e->d_ent_rec[l], (size_t)n_units * (P.unit_mb_cap + 2)))) return rc;
			if ((rc = dalloc(e, &e->d_pk_first[l], (size_t)n_units * P.pk_chunks_cap))) return rc;
			if ((rc = dalloc(e, &e->d_pk_agg[l], (size_t)n_units * P.pk_chunks_cap))) return rc;
			if ((rc = dalloc(e, &e->d_pk_unit[l], (size_t)n_units))) return rc;
		} else {
			if ((rc = dalloc(e, &e->d_rbsp[l], (size_t)n_units * P.rbsp_cap + 64))) return rc; // + one escape chunk of read slack
			if ((rc = dalloc(e, &e->d_esc[l], (size_t)n_units * P.esc_chunks_cap))) return rc;
		}
		if ((rc = dalloc(e, &e->d_nal_off[l], (size_t)n_units))) return rc;
		if ((rc = dalloc(e, &e->d_nal_size[l], (size_t)n_units))) return rc;
		if ((rc = dalloc(e, &e->d_stream_size[l], (size_t)MAX_COHORTS * (S + 4)))) return rc; // + k_copy_out's workgroup counter; per cohort
	}
	for (int i = 0; i < e->depth && !e->ent_fused; i++)
		if ((rc = dalloc(e, &e->d_out[i], e->out_cap + 16))) return rc; // k_copy_out moves whole 16-byte chunks
	return 0;
}

// What the plan chose for the two wavefronts, and per ring slot their sync words, hand-off buffers and progress words.
static int alloc_wavefront_words(cedar_enc *e, const Plan &plan)
{
	const int mb_w = e->P.mb_w, n_units = e->ext.n_streams * e->ext.slice_num;
	int rc = 0;
	e->wf_split = plan.wf_split;
	e->wf_split_intra = plan.wf_split_intra;
	e->db_pairs = plan.db_pairs;
	e->db_rows = plan.db_rows != 0;
	e->db_lds_kb = plan.db_lds_kb;
	e->db_bands = plan.db_bands;
	e->hb_slots = std::max(WF_MAX_SPLIT - 1, e->db_rows ? e->db_bands - 1 : 0);
	e->ir_rows = plan.ir_rows != 0;
	e->ir_lds_kb = plan.ir_lds_kb;
	for (int i = 0; i < e->depth; i++) {
		if ((rc = dalloc(e, &e->d_wf_sync[i], (size_t)MAX_COHORTS * 3 * WF_SYNC_WORDS))) return rc; // per cohort
		if ((rc = dalloc(e, &e->d_wf_hbuf[i], (size_t)2 * n_units * e->hb_slots * mb_w * WF_HBUF_GRANULES))) return rc; // deblock, intra
		if (e->db_rows) { // (tags decide what is this frame's: memory that held another session's must not look like it)
			const size_t n_hb = (size_t)n_units * e->db_bands * mb_w * DBR_HB_GRANULES;
			if ((rc = dalloc(e, &e->d_db_hbuf[i], n_hb))) return rc;
			HIP_TRY(e, hipMemset(e->d_db_hbuf[i], 0, n_hb * sizeof(uint64_t)));
		}
		if (e->ir_rows) { // (tagged like db_hbuf)
			const size_t n_hb = (size_t)n_units * 2 * e->db_bands * 4 * mb_w;
			if ((rc = dalloc(e, &e->d_ir_hbuf[i], n_hb))) return rc;
			HIP_TRY(e, hipMemset(e->d_ir_hbuf[i], 0, n_hb * sizeof(uint64_t)));
		}
		if ((rc = dalloc(e, &e->d_prog[i], (size_t)16 + 4 * n_units * PROG_BANDS))) return rc;
		HIP_TRY(e, hipMemset(e->d_prog[i], 0, sizeof(uint32_t) * (16 + 4 * (size_t)n_units * PROG_BANDS)));
	}
	return 0;
}

// The device side of the buffer manager: per ring slot the frame's planes, records, levels and hand-off words; per entropy
// lane the bit slots and what k_scan / k_pack_out meet in; the tables. Sizes follow the plan (frames in flight, which wavefront
// kernels, how many bands); everything is cleared (dalloc), tagged words on purpose (they never hold another frame's tag).
static int alloc_device_buffers(cedar_enc *e, const Plan &plan, int luma_sz, int chroma_sz, int G, bool row_rule)
{
	Params &P = e->P;
	const struct cedar_enc_config_ext &ext = e->ext;
	const int S = ext.n_streams, n_units = S * ext.slice_num, mb_w = P.mb_w, mb_h = P.mb_h;
	int rc = 0;
	(void)mb_h;
	const size_t plane_y = (size_t)P.cw * P.ch, plane_uv = plane_y / 2;
	P.plane_y_stride = (int64_t)plane_y;
	P.plane_uv_stride = (int64_t)plane_uv;
	// (GOP lanes take their frames from the mux's ring, never from these per-slot copies)
	if ((rc = dalloc(e, &e->d_in_y, (size_t)luma_sz * S * (G > 1 ? 1 : e->depth)))) return rc;
	if ((rc = dalloc(e, &e->d_in_uv, (size_t)chroma_sz * S * (G > 1 ? 1 : e->depth)))) return rc;
	if ((rc = dalloc(e, &e->d_in_ptrs, (size_t)2 * S * e->depth))) return rc;
	for (int i = 0; i < e->nrec; i++) {
		if ((rc = dalloc(e, &e->d_rec_y[i], plane_y * S))) return rc;
		if ((rc = dalloc(e, &e->d_rec_uv[i], plane_uv * S))) return rc;
	}
	for (int i = 0; i < e->depth; i++) {
		if ((rc = dalloc(e, &e->d_src_y[i], plane_y * S))) return rc;
		if ((rc = dalloc(e, &e->d_src_uv[i], plane_uv * S))) return rc;
		// + 1 macroblock: the dummy slot idle k_inter waves write to
		if ((rc = dalloc(e, &e->d_mbi[i], (size_t)P.n_mb * S + 1))) return rc;
		if ((rc = dalloc(e, &e->d_i4[i], ((size_t)P.n_mb * S + 1) * 16))) return rc;
		if ((rc = dalloc(e, &e->d_nnz[i], ((size_t)P.n_mb * S + 1) * NNZ_PER_MB))) return rc;
		if ((rc = dalloc(e, &e->d_coef[i], ((size_t)P.n_mb * S + 1) * COEF_PER_MB))) return rc;
		if (e->mb_qp) {
			if ((rc = dalloc(e, &e->d_qp_eff[i], (size_t)P.n_mb * S))) return rc;
			if ((rc = dalloc(e, &e->d_qp_t[i], (size_t)P.n_mb * S))) return rc;
		}
	}
	if (e->mb_qp && (rc = alloc_qp_tables(e, row_rule)))
		return rc;
	if ((rc = alloc_entropy_scratch(e)))
		return rc;
	if ((rc = dalloc(e, &e->d_pred4, (size_t)9 * 16))) return rc;
	if ((rc = dalloc(e, &e->d_abort_all, (size_t)16))) return rc;
	if (getenv("CEDAR_DEBUG_HANG") && (rc = dalloc(e, &e->d_dbg_log, (size_t)4 + 4 * DBG_LOG_ENTRIES))) return rc;
	for (int i = 0; i < e->depth && ext.scene_cut > 0; i++)
		if ((rc = dalloc(e, &e->d_cut_votes[i], (size_t)4))) return rc;
	if ((rc = alloc_wavefront_words(e, plan)))
		return rc;
	e->prev_slot = e->prev2_slot = -1;
	uint8_t tab[9 * 16];
	build_pred4_table(tab);
	HIP_TRY(e, hipMemcpy(e->d_pred4, tab, sizeof(tab), hipMemcpyHostToDevice));
	e->h_in_ptrs.assign((size_t)2 * S, nullptr);
	for (int s = 0; s < S; s++) {
		e->h_in_ptrs[s] = e->d_in_y + (size_t)s * luma_sz;
		e->h_in_ptrs[S + s] = e->d_in_uv + (size_t)s * chroma_sz;
	}
	for (bool &v : e->in_tab_valid)
		v = false;

	return 0;
}

// One pipelined session per device, across sessions and processes (see pipe_lock_acquire).
static void take_pipe_token(cedar_enc *e)
{
	if (e->holds_pipe_token)
		return;
	if (g_pipelined_sessions[e->device & 63].fetch_add(1) == 0) {
		e->pipe_lock_fd = pipe_lock_acquire(e->device);
		if (e->pipe_lock_fd == -1) // taken by a session of another process
			g_pipelined_sessions[e->device & 63].fetch_sub(1);
		else
			e->holds_pipe_token = true;
	} else {
		g_pipelined_sessions[e->device & 63].fetch_sub(1);
	}
}

// The deblock streams that really run beside s_main, the entropy stream and each other, moved to the front of s_db[]
// (at most db_max of them; *usable = how many).
static int probe_deblock_streams(cedar_enc *e, bool want, int db_max, int *usable_out)
{
	int usable = 0;
	for (int i = 0; want && i < DB_STREAMS && usable < db_max; i++) {
		bool ok = false, ok2 = true;
		if (Launch::probe_overlap(e->s_db[i], e->s_main, e->d_prog[0], &ok))
			return fail(e, -EIO, "stream probe failed");
		// ... and beside the entropy stream, or every frame's bitstream would queue behind a
		// deblock kernel that is resident for a whole frame time
		if (ok && (Launch::probe_overlap(e->s_db[i], e->s_ent, e->d_prog[0], &ok2) || !ok2))
			ok = false;
		ok2 = true;
		for (int j = 0; ok && j < usable; j++) // and beside the deblock streams already accepted
			if (Launch::probe_overlap(e->s_db[i], e->s_db[j], e->d_prog[0], &ok2) || !ok2)
				ok = false;
		if (ok) {
			hipStream_t t = e->s_db[usable];
			e->s_db[usable] = e->s_db[i];
			e->s_db[i] = t;
			usable++;
		}
	}
	HIP_TRY(e, hipMemset(e->d_prog[0], 0, 64));
	HIP_TRY(e, hipStreamSynchronize(nullptr)); // (the null stream's clear before the probes' own streams touch the words)
	*usable_out = usable;
	return 0;
}

static int probe_analysis_streams(cedar_enc *e)
{
	// further analysis streams: each must run beside every deblock stream and every analysis
	// stream already accepted (a frame's gate kernels spin while the next frame's run)
	int an = 1;
	const char *enva = getenv("CEDAR_ANALYSIS_STREAMS");
	const int an_max = enva ? atoi(enva) : 1;
	for (int i = 1; e->row_pipe && i < AN_STREAMS && an < an_max; i++) {
		bool ok = true, o = false;
		for (int j = 0; ok && j < e->db_streams; j++)
			if (Launch::probe_overlap(e->s_ana[i], e->s_db[j], e->d_prog[0], &o) || !o)
				ok = false;
		for (int j = 0; ok && j < an; j++)
			if (Launch::probe_overlap(e->s_ana[i], e->s_ana[j], e->d_prog[0], &o) || !o)
				ok = false;
		if (ok) {
			hipStream_t t = e->s_ana[an];
			e->s_ana[an] = e->s_ana[i];
			e->s_ana[i] = t;
			an++;
		}
	}
	e->an_streams = an;
	return 0;
}

static int setup_lookahead(cedar_enc *e, const struct cedar_enc_config *cfg, const Plan &plan)
{
	// Look-ahead for IDR pictures: two more streams -- the IDR picture's analysis (busy for
	// milliseconds) and the input of all frames (which must never queue behind that, nor behind
	// a gate on s_main) -- dealt by assign_stream_roles; each must run beside every stream in use.
	// The IDR picture's deblocking: on a stream of its own, resident beside the intra wavefront it follows band by
	// band (k_deblock_rows, GOPs of 20 pictures and more: 1080p keyint 25, K 300 47.0 -> 48.0 M MB/s, the driver's K 20
	// 34.8 -> 38.3 M: its 0.3 ms no longer come after the intra kernel's 1.9 ms), or behind it on the same stream --
	// k_deblock2_planes, whose whole-CU workgroups measured slower beside it (one more queue that blocks a pipe), and
	// SHORT GOPs on either kernel: the stream that would carry it carries every other IDR picture instead (s_idr2),
	// and a stream of short GOPs is bound by its IDR pictures (keyint 1 / 4 / 8 / 12: 7.2 / 25.1 / 44.5 / 46.2 M MB/s
	// behind against 4.1 / 15.1 / 29.3 / 42.2 beside). CEDAR_IDR_DEBLOCK_INLINE.
	const char *envl = getenv("CEDAR_LOOKAHEAD"), *envi = getenv("CEDAR_IDR_DEBLOCK_INLINE");
	e->lookahead = false;
	e->idr_deblock_inline = envi ? atoi(envi) != 0 : !e->db_rows || cfg->keyframe_interval < 20 || !plan.idr_beside_ok;
	// (sessions whose frames run back to back take the IDR picture out of line too: nothing polls
	// there, the P picture after it waits for its events, and with 4 frames in flight it gets a
	// three-frame head start)
	const char *envlb = getenv("CEDAR_LOOKAHEAD_BATCH");
	if ((e->row_pipe ? e->depth > 4 : ((!envlb || atoi(envlb) != 0) && e->depth >= 3)) && (!envl || atoi(envl) != 0)) {
		// streams not taken by the deblock / analysis roles, in the order assign_stream_roles meant
		// them for: IDR analysis, input, then the spare one (second entropy lane)
		std::vector<hipStream_t> used = { e->s_main, e->s_ent }, spare, got;
		for (int i = 0; i < e->db_streams; i++)
			used.push_back(e->s_db[i]);
		for (int i = 1; i < e->an_streams; i++)
			used.push_back(e->s_ana[i]);
		for (int i = e->an_streams; i < AN_STREAMS; i++)
			spare.push_back(e->s_ana[i]);
		for (int i = e->db_streams; i < DB_STREAMS; i++)
			spare.push_back(e->s_db[i]);
		const size_t want = e->idr_deblock_inline ? 3 : 4;
		for (size_t c = 0; c < spare.size() && got.size() < want; c++) {
			bool ok = true, o = false;
			for (size_t j = 0; ok && j < used.size(); j++)
				if (Launch::probe_overlap(spare[c], used[j], e->d_prog[0], &o) || !o)
					ok = false;
			if (ok) {
				got.push_back(spare[c]);
				used.push_back(spare[c]);
			}
		}
		HIP_TRY(e, hipMemset(e->d_prog[0], 0, 64));
		HIP_TRY(e, hipStreamSynchronize(nullptr)); // (the null stream's clear before the probes' own streams touch the words)
		if (got.size() >= (e->idr_deblock_inline ? 2u : 3u)) {
			e->s_idr = got[0];
			e->s_in = got[1];
			size_t k = 2;
			e->s_db_idr = e->idr_deblock_inline ? nullptr : got[k++];
			const char *envl2 = getenv("CEDAR_ENTROPY_LANES");
			e->s_ent2 = k < got.size() && e->ent_lanes > 1 && (!envl2 || atoi(envl2) > 1) ? got[k] : nullptr;
			// the spare stream otherwise carries every other IDR picture: consecutive IDR pictures
			// depend on nothing either, which doubles the rate of short GOPs (keyint 1: all of them)
			e->s_idr2 = !e->s_ent2 && e->idr_deblock_inline && k < got.size() ? got[k] : nullptr;
			e->lookahead = true;
		}
	}
	return 0;
}

static int setup_cohorts(cedar_enc *e, const PlanRun &run)
{
	// cohorts (submit_frame): two from 32 units on. With whole-CU deblock workgroups the deblocking
	// of one cohort never ran beside the search of the other (16 waves of 128 registers find no
	// empty CU while k_inter's small workgroups keep refilling them: 32 x 1080p 2.29 ms per step
	// against 2.20 ms with one cohort; CU masks 2.96 ms; stream priority no change). With the
	// 2-pair workgroups it does: 32 x 1080p 1.87 -> 1.73 ms (139 -> 151 M MB/s), 64: 3.58 -> 3.18 ms
	// (146 -> 164 M). Not below: the deblock wavefront of a picture takes its ~0.45 ms however few
	// pictures there are, and half of 16 streams' search is shorter than that (16: 120 -> 92 M).
	// Three or four cohorts lose everywhere (more, smaller launches on the same four hardware queues).
	// Wave priority (s_setprio) for the deblock waves and sleeping instead of spinning in their
	// waits changed nothing either way.
	const int C = run.cohorts;
	e->cohorts = C;
	// CEDAR_COHORT_STREAMS (development): bit 0 the deblocking, bit 1 the entropy chain of cohort c > 0 on a
	// stream of its own. Off by default: measured at 32 x 1080p, both on 162.7 M MB/s against 168.9 M on the
	// shared streams -- more queues holding resident kernels slow each other's launches.
	const char *envs = getenv("CEDAR_COHORT_STREAMS"), *envr = getenv("CEDAR_BATCH_SELF_RESET");
	const int cs = envs ? atoi(envs) : 0;
	e->batch_self_reset = envr && atoi(envr) != 0;
	if (C > 1 && cs) {
		for (int c = 1; c < C; c++) {
			if ((cs & 1) && c < DB_STREAMS)
				e->s_db_c[c] = e->s_db[DB_STREAMS - c]; // (the session's spare deblock streams, from the far end: s_db[1] may carry every other IDR picture)
			if (cs & 2)
				HIP_TRY(e, hipStreamCreateWithFlags(&e->s_ent_c[c], hipStreamNonBlocking));
		}
		if (cs & 2)
			for (int i = 0; i < e->depth * MAX_COHORTS; i++)
				HIP_TRY(e, hipEventCreateWithFlags(&e->ev_entropy[i], hipEventDisableTiming));
	}
	return 0;
}

// The session's streams and what the probes make of the plan: which deblock streams really run beside the analysis and the
// entropy stream (frame pipelining polls across them), which hardware queues share a pipe (the roles are dealt by that), the
// IDR look-ahead streams, cohorts.
static int setup_streams(cedar_enc *e, const struct cedar_enc_config *cfg, const PlanIn &pin, const Plan &plan)
{
	Params &P = e->P;
	(void)P;
	(void)cfg;
	// CEDAR_STREAM_PRIO=1: the first stream of the pool at the greatest stream priority. For callers whose process keeps its
	// NULL stream busy (PyTorch's default stream): with 7 streams and the null stream on 8 hardware queues, the stream the
	// roles give the analysis -- the one without a pipe-mate among the session's seven (assign_stream_roles) -- has the null
	// stream's queue for its mate, and a null stream that holds a running kernel with a packet behind it costs every launch
	// of the analysis stream ~20 us: one 1080p stream 60 -> 19 M MB/s with spinning kernels queued there, 59.8 M with this
	// switch (tools/dev/null_blocked.py, profiles/r04_null_stream.txt). The roles are then dealt in classes -- analysis and
	// entropy on the first two streams of the pool (the analysis on the one with the priority, which its own mate no longer
	// slows), the deblock roles on the next three, IDR and input on the last two: the null stream's mate, the fourth stream
	// in creation order, gets a deblock role, one or two launches a frame. Off by default: a queue of another priority in
	// the process moved other sessions of the same process (854x480 17.5 -> 15 M; 32 streams with it themselves 196 -> 169).
	{
		const char *envp = getenv("CEDAR_STREAM_PRIO");
		int pr_least = 0, pr_greatest = 0;
		e->main_prio = envp && atoi(envp) != 0 && hipDeviceGetStreamPriorityRange(&pr_least, &pr_greatest) == hipSuccess && pr_least != pr_greatest;
		if (e->main_prio)
			HIP_TRY(e, hipStreamCreateWithPriority(&e->s_main, hipStreamNonBlocking, pr_greatest));
		else
			HIP_TRY(e, hipStreamCreateWithFlags(&e->s_main, hipStreamNonBlocking));
	}
	e->s_ana[0] = e->s_main;
	HIP_TRY(e, hipStreamCreateWithFlags(&e->s_ent, hipStreamNonBlocking));
	for (int i = 0; i < DB_STREAMS; i++)
		HIP_TRY(e, hipStreamCreateWithFlags(&e->s_db[i], hipStreamNonBlocking));
	for (int i = 1; i < AN_STREAMS; i++)
		HIP_TRY(e, hipStreamCreateWithFlags(&e->s_ana[i], hipStreamNonBlocking));
	if (e->depth > 4) {
		const int rc_roles = assign_stream_roles(e, plan.db_streams_wanted);
		if (rc_roles)
			return rc_roles;
	}
	e->split_planes = plan.split_planes != 0;
	HIP_TRY(e, hipEventCreateWithFlags(&e->ev_host_in, hipEventDisableTiming));
	if (getenv("CEDAR_TRACE_TIMELINE")) {
		HIP_TRY(e, hipEventCreate(&e->ev_base));
		HIP_TRY(e, hipEventRecord(e->ev_base, e->s_main));
		HIP_TRY(e, hipEventSynchronize(e->ev_base));
	}
	{
		// Frame pipelining needs kernels of s_main and of the deblock streams to be on the GPU at
		// the same time (they poll each other's progress words). Probe it: a stream that shares
		// its hardware queue with s_main would wait forever (until the spin bound) otherwise.
		// And it must leave the GPU mostly free: a pipelined deblock kernel is resident (one CU per
		// workgroup, nothing else fits beside it) while it waits for its frame's analysis, so the
		// deblock workgroups of all frames in flight may only take a fraction of the CUs, or the
		// analysis they wait for finds no CU to run on.
		const int db_max = plan.db_streams_wanted;
		bool want = plan.want_row_pipe != 0;
		if (want)
			take_pipe_token(e);
		want = want && e->holds_pipe_token;
		int usable = 0;
		if (int rc_p = probe_deblock_streams(e, want, db_max, &usable))
			return rc_p;
		const PlanRun run = plan_run(pin, plan, PlanProbe{ usable, e->holds_pipe_token ? 1 : 0 });
		e->row_pipe = run.row_pipe != 0;
		e->db_streams = run.db_streams;
		if (int rc_p = probe_analysis_streams(e))
			return rc_p;
		e->inter_bands = run.inter_bands;
		e->defer_default = run.defer_default;
		if (int rc_p = setup_lookahead(e, cfg, plan))
			return rc_p;
		if (int rc_p = setup_cohorts(e, run))
			return rc_p;
		if (getenv("CEDAR_TRACE_SUBMIT"))
			fprintf(stderr, "config: row_pipe %d, deblock streams %d, analysis streams %d, stripes %d, look-ahead %d, depth %d, cohorts %d\n",
				(int)e->row_pipe, e->db_streams, e->an_streams, e->inter_bands, (int)e->lookahead, e->depth, e->cohorts);
	}
	return 0;
}

// Steps of k_deblock_rows between two progress words of a band. Every word costs its writer wave a release fence (an L2
// write-back beside a busy chip), every step of delay costs the gates of the next picture's stripes: measured at the end of
// round 4 (tools/dev/r04_single_sweep{6,7}.txt, tools/dev/pub_check.sh; K 300, M MB/s) 2 / 4 / 8 / 16 / 32 steps: 49.3 / 56.8 /
// 59-60 / 60.8-61.2 / 60.1-60.3 for one 1080p stream, 16 against 8: 3840x2160 91-93 against 83-86, in four slices 93 against
// 88-90, two 1080p streams 85 against 83.3. CEDAR_DB_PUB_STEP (development).
static int db_pub_step()
{
	const char *envp = getenv("CEDAR_DB_PUB_STEP");
	const int st = envp ? atoi(envp) : 16;
	return st == 2 || st == 4 || st == 8 || st == 32 ? st : 16;
}

// The diagonal stripes of a pipelined P picture's analysis and how many of them are held back (DeferredTail).
static int setup_stripes(cedar_enc *e)
{
	Params &P = e->P;
	const struct cedar_enc_config_ext &ext = e->ext;
	const int mb_w = P.mb_w, mb_h = P.mb_h;
	int rc = 0;
	(void)ext;
	(void)mb_w;
	(void)mb_h;
	(void)rc;
	{
		// diagonal stripes s = x + y of a P picture (the slope of the deblock sweep it follows and is
		// followed by) and, per stripe, how many of its macroblocks lie in the rows before row r: k_inter
		// enumerates a stripe without holes
		const int nb = e->inter_bands;
		const int d_lo = P.row_first, d_hi = P.mb_w - 1 + (P.row_first + P.row_num - 1) + 1;
		e->stripe_s.assign((size_t)nb + 1, 0);
		e->stripe_n.assign((size_t)nb, 0);
		std::vector<uint32_t> tab((size_t)nb * (P.row_num + 1), 0);
		for (int k = 0; k <= nb; k++)
			e->stripe_s[k] = d_lo + (int)((long)(d_hi - d_lo) * k / nb);
		for (int k = 0; k < nb; k++) {
			uint32_t run = 0;
			for (int r = 0; r <= P.row_num; r++) {
				tab[(size_t)k * (P.row_num + 1) + r] = run;
				const int y = P.row_first + r;
				auto cl = [&](int v) { return v < 0 ? 0 : v > P.mb_w ? P.mb_w : v; };
				run += (uint32_t)(cl(e->stripe_s[k + 1] - y) - cl(e->stripe_s[k] - y));
			}
			e->stripe_n[k] = (int)tab[(size_t)k * (P.row_num + 1) + P.row_num];
		}
		// Which stripes may be held back (see DeferredTail). Own stripe j of the next picture is enqueued
		// after this picture's stripes up to defer_from + j - 1; its gate waits for this picture's
		// deblocking to have ALL bands at x + y >= s1(j) + 4, i.e. its top band nb - 1 + 5 steps further,
		// plus the 8 steps between two publications, and that deblocking asks for its own analysis up to
		// one diagonal more: everything below s1(j) + nb + 14 must already be in the queue, or the stream
		// waits for itself.
		{
			const char *envd = getenv("CEDAR_DEFER");
			int bands_max = 0;
			for (int s2 = ext.slice_first; s2 < ext.slice_first + ext.slice_num; s2++) {
				int a, b;
				slice_rows(P.mb_h, P.slice_count, s2, &a, &b);
				bands_max = std::max(bands_max, (b - a + 3) / 4);
			}
			// (all of it assumes the deblock sweep of a unit is ONE front, every band 5 steps behind the band
			// above: a workgroup with more bands than wave pairs starts band b + pairs when band b is through)
			const bool one_front = e->db_rows || (bands_max + e->wf_split - 1) / e->wf_split <= e->db_pairs; // (a workgroup per band: always)
			// CEDAR_DEFER = number of stripes to hold back (0: none). Between two submits the device then holds
			// kernels that wait for work only the NEXT submit enqueues; a device-wide synchronisation by anybody in
			// the process in between would wait for them, so the session's flush thread enqueues the tail itself
			// CEDAR_DEFER_FLUSH_US (100) after the submit unless the next submit has come (Flusher).
			// tail_gate (a stripe's launch ends when the next stripe's gate would have passed): only with the stripes of a picture
			// one behind the other on the stream -- a held-back stripe of the picture before, enqueued between two of them, would
			// sit behind the very wait it is needed for. CEDAR_TAIL_GATE=0: a k_gate in front of every stripe, as until round 4.
			e->tail_gate = e->row_pipe && e->cohorts == 1 && !(getenv("CEDAR_TAIL_GATE") && atoi(getenv("CEDAR_TAIL_GATE")) == 0);
			const int want = e->tail_gate ? 0 : envd ? atoi(envd) : e->defer_default;
			int kd = nb;
			if (e->row_pipe && e->cohorts == 1 && !e->mux && want > 0 && want < nb && nb >= 3 && one_front)
				kd = nb - want;
			// Safe only if the previous picture's deblocking can get as far as this picture's stripe j asks (k_gate:
			// diagonal stripe_s[j + 1] + 4, in every band that has such diagonals) on what is published of the
			// previous picture's analysis when that gate runs: its stripes up to kd + j - 1 are in front of the gate,
			// but the end of a stripe is published by the gate of the NEXT one, so diagonal stripe_s[kd + j - 1] it is.
			// How far ahead of the asked diagonal the analysis must be: the progress words move every db_pub_step() steps, a
			// stager waits for the analysis of 5 steps ahead (+5), and -- k_deblock_rows -- every band is on the SAME
			// diagonal as the band above (one macroblock behind its last row, like the rows of a band). (k_deblock2_planes,
			// bands inside one workgroup, every band a diagonal behind: the margin it has always run with.)
			const int slack = e->db_rows ? 4 + db_pub_step() + 5 + 3 : bands_max + 14;
			for (int j = 0; j < kd && kd < nb; j++)
				if (e->stripe_s[j + 1] + slack > e->stripe_s[std::min(nb, kd + j - (e->db_rows ? 1 : 0))])
					kd = nb;
			e->defer_from = kd;
			if (kd < nb) {
				const char *envu = getenv("CEDAR_DEFER_FLUSH_US");
				e->flusher = new Flusher();
				e->flusher->us = envu && atol(envu) > 0 ? atol(envu) : 100;
				e->flusher->th = std::thread(flusher_main, e);
			}
		}
		if (e->d_stripe_tab) { hipFree(e->d_stripe_tab); e->d_stripe_tab = nullptr; }
		if ((rc = dalloc(e, &e->d_stripe_tab, tab.size()))) return rc;
		HIP_TRY(e, hipMemcpy(e->d_stripe_tab, tab.data(), tab.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
		if (!e->d_dummy_rec && (rc = dalloc(e, &e->d_dummy_rec, (size_t)24 * P.cw))) return rc;
		e->P.dummy_rec = e->d_dummy_rec;
		e->P.stripe_tab = nullptr;
	}
	return 0;
}

// the events of every device slot: input there, analysis complete, reconstruction deblocked (per cohort), the stage timers
static int create_events(cedar_enc *e)
{
	for (int i = 0; i < e->depth; i++) {
		HIP_TRY(e, hipEventCreateWithFlags(&e->fl[i].done_main, hipEventDisableTiming));
		for (int c = 0; c < e->cohorts; c++) {
			HIP_TRY(e, hipEventCreateWithFlags(&e->ev_input[i * MAX_COHORTS + c], hipEventDisableTiming));
			HIP_TRY(e, hipEventCreateWithFlags(&e->ev_analysis[i * MAX_COHORTS + c], hipEventDisableTiming));
			HIP_TRY(e, hipEventCreateWithFlags(&e->ev_deblocked[i * MAX_COHORTS + c], hipEventDisableTiming));
		}
		HIP_TRY(e, hipEventCreateWithFlags(&e->fl[i].done_ent, hipEventDisableTiming));
		for (int s = 0; s < CEDAR_ENC_STAGE_COUNT; s++)
			for (int k = 0; k < 2; k++)
				HIP_TRY(e, hipEventCreate(&e->fl[i].tm.ev[s][k]));
	}
	return 0;
}

// everything of Params that is a device pointer or a choice of the plan: what submit_frame copies and then points at the
// frame's own slot
static void static_pointers(cedar_enc *e, int n_units, int mb_w)
{
	Params &P = e->P;
	const int S = e->ext.n_streams;
	P.in_luma = (const uint8_t *const *)e->d_in_ptrs;
	P.in_chroma = (const uint8_t *const *)(e->d_in_ptrs + S);
	P.src_y = e->d_src_y[0];
	P.src_uv = e->d_src_uv[0];
	P.slot = e->d_slot[0];
	P.mb_bits = e->d_mb_bits[0];
	P.skip = e->d_skip[0];
	P.mvd = nullptr;
	P.bit_off = e->d_bit_off[0];
	P.run_before = e->d_run[0];
	P.src_direct = 0;
	P.src_last_y = P.ch - 1;
	P.src_last_c = P.ch / 2 - 1;
	P.ent_rec = e->d_ent_rec[0];
	P.pk_first = e->d_pk_first[0];
	P.pk_agg = e->d_pk_agg[0];
	P.pk_unit = e->d_pk_unit[0];
	P.unit_bits = e->d_unit_bits[0];
	P.rbsp = e->d_rbsp[0];
	P.esc_cnt = e->d_esc[0];
	P.nal_off = e->d_nal_off[0];
	P.nal_size = e->d_nal_size[0];
	P.stream_size = e->d_stream_size[0];
	P.out = e->d_out[0];
	P.hdr = nullptr; // per frame: the frame's own pinned header words
	P.prefix_len = nullptr;
	P.pred4_tab = e->d_pred4;
	P.abort_all = e->d_abort_all;
	P.dbg_log = e->d_dbg_log;
	P.wf_sync = e->d_wf_sync[0];
	P.wf_hbuf = e->d_wf_hbuf[0];
	P.db_hbuf = e->d_db_hbuf[0];
	P.db_tag = 0;
	P.wf_hbuf_intra = e->d_wf_hbuf[0] + (size_t)n_units * e->hb_slots * mb_w * WF_HBUF_GRANULES;
	P.prog_inter = nullptr;
	P.prog_db = nullptr;
	P.prog_intra = nullptr;
	P.prog_intra_c = nullptr;
	P.prev_db = nullptr;
	P.stripe_s0 = 0;
	P.stripe_s1 = 0;
	P.prog_tag = P.prev_tag = 0;
	P.wf_split = e->wf_split;
	P.db_pairs = e->db_pairs;
	P.wf_hb_slots = e->hb_slots;
	P.db_rows = e->db_rows ? 1 : 0;
	P.db_pub_mask = db_pub_step() - 1;
	P.gate_publish = -1;
	P.gate_next_need = 0;
	P.db_lds_kb = getenv("CEDAR_DEBLOCK_LDS_KB") ? std::max(1, atoi(getenv("CEDAR_DEBLOCK_LDS_KB"))) : e->db_lds_kb; // (see launch_deblock_rows)
	P.db_bands = e->db_bands;
	P.wf_split_intra = e->wf_split_intra;
	P.ir_rows = e->ir_rows ? 1 : 0;
	P.ir_bands = e->db_bands;
	P.ir_lds_kb = getenv("CEDAR_INTRA_LDS_KB") ? std::max(0, atoi(getenv("CEDAR_INTRA_LDS_KB"))) : e->ir_lds_kb; // (see launch_intra_rows)
	P.ir_hbuf = e->d_ir_hbuf[0];
	{
		const char *env = getenv("CEDAR_FAULT_INJECT"); // tests: lose a hand-off, see that the grid drains and collect says -EIO
		P.fault_inject = env ? atoi(env) : 0;
		env = getenv("CEDAR_FAULT_INJECT_AT"); // ... of the n-th submit only (1-based; default: of every frame)
		e->fault_at = env ? (uint32_t)atoi(env) : 0;
	}
}

// gop_parallel: the rings between the caller's frame order and the lock-step lanes (GopMux)
static int setup_gop_lanes(cedar_enc *e, const struct cedar_enc_config *cfg, int G, int S_user)
{
	const struct cedar_enc_config_ext &ext = e->ext;
	GopMux *m = new (std::nothrow) GopMux();
	if (!m)
		return fail(e, -ENOMEM, "out of memory");
	e->mux = m;
	m->G = G;
	m->K = cfg->keyframe_interval;
	m->S = S_user;
	// frames the caller may have in flight: (G-1) K is the least (frame n needs frame n + (G-1) K in),
	// + the session's ring; with `ahead` another (G-1) K + 8, so that the IDR step of the NEXT group of
	// GOPs has its input while this group's last steps still run and hides beside them (the session's
	// IDR look-ahead; measured below) -- at the price of twice the latency
	m->ahead = !ext.gop_low_latency;
	m->user_depth = (G - 1) * m->K + e->depth + (m->ahead ? (G - 1) * m->K + 8 : 0);
	m->cap = m->user_depth + m->K + 1;
	m->in_ring.assign((size_t)m->cap * 2 * S_user, nullptr);
	m->out_ring.resize((size_t)m->cap * S_user);
	m->out_rc.assign((size_t)m->cap, INT32_MIN);
	m->user_in.assign((size_t)2 * S_user, nullptr);
	return 0;
}

int config_impl(cedar_enc *e, struct cedar_enc_config *cfg, const struct cedar_enc_config_ext *ext_in)
{
	struct cedar_enc_config_ext ext{};
	if (ext_in) {
		if (ext_in->size != (int)sizeof(struct cedar_enc_config_ext))
			return fail(e, -EINVAL, "config_ext.size mismatch");
		ext = *ext_in;
	}
	if (ext.n_streams == 0) ext.n_streams = 1;
	if (ext.gop_parallel < 0 || ext.gop_parallel > MAX_UNITS || ext.n_streams < 1 || ext.n_streams > MAX_UNITS)
		return fail(e, -EINVAL, "bad stream/slice layout");
	// the caller's streams x GOP lanes = the lock-step "streams" everything below batches
	const int S_user = ext.n_streams, G = ext.gop_parallel > 1 ? ext.gop_parallel : 1;
	ext.n_streams = S_user * G;
	if (ext.slice_count == 0) ext.slice_count = 1;
	if (ext.slice_num == 0) { ext.slice_first = 0; ext.slice_num = ext.slice_count; }
	ext.search_range = ext.search_range == 0 ? 16 : ext.search_range < 0 ? 0 : ext.search_range;
	e->chroma_qp_offset = 4 + ext.chroma_qp_offset_minus4; // cedar.c:971-972
	{
		const int rc_v = validate_config(e, cfg, ext, S_user, G);
		if (rc_v)
			return rc_v;
	}
	const int mb_h = cfg->dst_height / 16, mb_w = cfg->dst_width / 16;
	(void)mb_h;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ext.device < 0 || ext.device >= ndev)
		return fail(e, -ENODEV, "HIP device %d not available", ext.device);
	HIP_TRY(e, hipSetDevice(ext.device));
	e->device = ext.device;
	e->device_set = true;
	e->cfg = *cfg;
	e->ext = ext;
	e->mb_qp = ext.roi_count > 0;
	for (int k = 0; k < 5; k++)
		e->mb_qp = e->mb_qp || ext.mad_dqp[k] != 0;
	const bool row_rule = e->mb_qp;
	// rate control over several streams: every stream its own QP per picture, i.e. the kernels that look the
	// QP up per macroblock (but no QP differs inside a picture: no row rule)
	e->rc_multi = ext.bitrate > 0 && ext.n_streams > 1 && G == 1;
	e->rc_lanes = ext.bitrate > 0 && G > 1; // one QP per group of G GOPs (rc_pick_qp_group)
	e->mb_qp = e->mb_qp || e->rc_multi;
	e->pcm_session = ext.pcm >= 0 && !e->mb_qp && ext.intra_refresh == 0;
	if (e->rc_multi) {
		e->rc_bits_s.assign((size_t)ext.n_streams * 64, 0);
		e->rc_qp_s.assign((size_t)ext.n_streams * 64, 0);
	}
	fill_static_params(e);
	Params &P = e->P;
	const int S = ext.n_streams;
	const int n_units = S * ext.slice_num;

	// The schedule policy is a pure function of geometry, CU count and the CEDAR_* switches (enc_plan.h; held
	// by tests/test_plan.py without a GPU): frames in flight, wavefront splits, deblock form, what to probe for.
	static_assert(CEDAR_PLAN_MAX_DEPTH == CEDAR_ENC_PIPELINE_DEPTH, "enc_plan.h");
	int cus = 0;
	HIP_TRY(e, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->device));
	PlanIn pin = plan_input(P.mb_w, P.mb_h, P.slice_count, ext.slice_first, ext.slice_num, S, cus);
	pin.mb_qp = e->mb_qp ? 1 : 0;
	const Plan plan = plan_static(pin);
	if (plan.error)
		return fail(e, -EINVAL, "picture too tall for the deblock wavefront");
	e->depth = plan.depth;
	e->nrec = e->depth + 1;
	// ---- buffer manager (cedar.c:605-704) ----
	const int luma_sz = align_up(cfg->src_width * cfg->src_height, 4096);      // cedar.c:612
	const int chroma_sz = align_up(cfg->src_width * cfg->src_height / 2, 4096); // cedar.c:621
	// reference: fixed 1 MiB (cedar.c:663), which a 1080p IDR can overflow; size by macroblocks
	int bs_sz = P.n_mb * 1024 + 4096;
	if (bs_sz < (1 << 20)) bs_sz = 1 << 20;
	e->out_cap = (size_t)bs_sz * S;
	e->stream_out_cap = (size_t)bs_sz;
	if (e->out_cap > 0x7fffffffULL)
		return fail(e, -EINVAL, "too many streams for one session");
	int rc = 0;
	if ((rc = alloc_host_buffers(e, luma_sz, chroma_sz, S_user, S, n_units)))
		return rc;
	if ((rc = alloc_device_buffers(e, plan, luma_sz, chroma_sz, G, row_rule)))
		return rc;

	if ((rc = setup_streams(e, cfg, pin, plan)))
		return rc;
	if ((rc = setup_stripes(e)))
		return rc;
	if ((rc = create_events(e)))
		return rc;
	HIP_TRY(e, hipDeviceSynchronize());

	static_pointers(e, n_units, mb_w);

	e->rc_target = ext.bitrate > 0 ? (int64_t)ext.bitrate * (ext.fps_num > 0 ? ext.fps_den : 1) / (ext.fps_num > 0 ? ext.fps_num : 30) : 0;
	build_sps_pps(e);
	e->frame_count = 0; // cedar.c:815-816
	e->frame_p_count = 0;
	e->res_off.assign(S, 0);
	e->res_size.assign(S, 0);
	e->lane_gidx.assign(S, 0);
	e->S_user = S_user;
	if (G > 1 && (rc = setup_gop_lanes(e, cfg, G, S_user)))
		return rc;
	e->configured = true;

	// outputs (cedar.c:840-860), host pointers instead of DMA addresses
	cfg->input_luma = e->h_luma;
	cfg->input_luma_size = luma_sz;
	cfg->input_chroma = e->h_chroma;
	cfg->input_chroma_size = chroma_sz;
	cfg->bytestream = e->h_out[0];
	cfg->bytestream_size = bs_sz;
	cfg->thumbnail = 0;
	cfg->thumb_luma = nullptr;
	cfg->thumb_luma_size = 0;
	cfg->thumb_chroma = nullptr;
	cfg->thumb_chroma_size = 0;
	e->cfg = *cfg;
	return 0;
}

} // namespace cedar

extern "C" int cedar_enc_plan(const struct cedar_enc_plan_query *q, struct cedar_enc_plan_result *r)
{
	if (!q || !r || q->mb_w < 1 || q->mb_h < 1 || q->slice_count < 1 || q->slice_count > q->mb_h || q->n_streams < 1 || q->cus < 1 ||
	    q->slice_first < 0 || q->slice_num < 0 || q->slice_first + q->slice_num > q->slice_count)
		return -EINVAL;
	const int num = q->slice_num ? q->slice_num : q->slice_count - q->slice_first;
	const PlanIn in = plan_input(q->mb_w, q->mb_h, q->slice_count, q->slice_first, num, q->n_streams, q->cus);
	const Plan p = plan_static(in);
	const PlanRun run = plan_run(in, p, PlanProbe{ q->db_streams_usable, q->holds_pipe_token });
	*r = cedar_enc_plan_result{ p.error, p.depth, p.wf_split, p.wf_split_intra, p.db_pairs, p.split_planes, p.want_row_pipe,
				    p.db_streams_wanted, run.row_pipe, run.db_streams, run.inter_bands, run.defer_default, run.cohorts, p.db_rows,
				    p.ir_rows, (2 * p.n_units * p.db_bands + 7) / 8, p.db_lds_kb, p.idr_beside_ok };
	return 0;
}
