// cedarx_h264_encoder_amd/csrc/enc_session.cpp
// Host side of the C-ABI (include/cedar_h264enc.h): what the reference does in
// its ioctl handlers, minus the register poking.
//   config validation + buffer manager   kernel/cedar.c:732-866, :605-704
//   header bit-writer (SPS/PPS/slice)    kernel/cedar.c:187-223, :868-1030
//   GOP counters, reference ping-pong    kernel/cedar.c:1047-1050, :1193-1201
//   encode = program + trigger + wait    kernel/cedar.c:1032-1209
// The "trigger + wait for IRQ" (cedar.c:1176-1180) becomes kernel launches on
// two HIP streams (analysis/deblock and entropy) and an event wait.
#include "enc_session.h"

namespace cedar {

int fail(cedar_enc *e, int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(e->err, sizeof(e->err), fmt, ap);
	va_end(ap);
	return code;
}

} // namespace cedar

// ===============================================================// C-ABI
// ===============================================================
extern "C" const char *cedar_enc_version(void)
{
	return "cedar_h264enc 0.1 (gfx950, HIP)";
}

extern "C" cedar_enc *cedar_enc_open(void)
{
	// More hardware queues than the runtime's default 4 (takes effect if HIP has not started yet
	// in this process; the session probes what it really got and adapts)
	setenv("GPU_MAX_HW_QUEUES", "8", 0);
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
		errno = ENODEV; // no CPU fallback: the encode path exists only as HIP kernels
		return nullptr;
	}
	cedar_enc *e = new (std::nothrow) cedar_enc();
	if (!e)
		errno = ENOMEM;
	return e;
}

extern "C" const char *cedar_enc_last_error(cedar_enc *e)
{
	return e ? e->err : "null session";
}

extern "C" int cedar_enc_config(cedar_enc *e, struct cedar_enc_config *cfg)
{
	return cedar_enc_config_ext(e, cfg, nullptr);
}

extern "C" int cedar_enc_config_ext(cedar_enc *e, struct cedar_enc_config *cfg, const struct cedar_enc_config_ext *ext_in)
{
	if (!e)
		return -EINVAL;
	if (!cfg) // cedar.c:738-739
		return fail(e, -EINVAL, "NULL config");
	if (e->configured) // cedar.c:744-747
		return fail(e, -EINVAL, "already configured");
	const int rc = config_impl(e, cfg, ext_in);
	if (rc) {
		// a CONFIG that failed half way leaves nothing behind (cedar.c:834-836 frees what
		// cedar_buffers_init had got): the session is as freshly opened and may be configured again
		char msg[sizeof(e->err)];
		memcpy(msg, e->err, sizeof(msg));
		if (e->device_set) // (a parameter refused before that must not touch whatever device the caller has current)
			hipDeviceSynchronize();
		pipe_token_release(e->device, e->holds_pipe_token, e->pipe_lock_fd);
		free_all(e);
		*e = cedar_enc();
		memcpy(e->err, msg, sizeof(msg));
	}
	return rc;
}


extern "C" int cedar_enc_set_device_input(cedar_enc *e, int stream, const void *d_luma, const void *d_chroma)
{
	if (!e || !e->configured || stream < 0 || stream >= e->S_user)
		return -EINVAL;
	if (e->mux) { // the frame is filed under its index at submit; lanes get their pointers per step
		const bool dev = d_luma && d_chroma;
		e->mux->user_in[stream] = dev ? (const uint8_t *)d_luma : nullptr;
		e->mux->user_in[e->S_user + stream] = dev ? (const uint8_t *)d_chroma : nullptr;
		return 0;
	}
	const int S = e->ext.n_streams;
	if (d_luma && d_chroma) {
		e->h_in_ptrs[stream] = (const uint8_t *)d_luma;
		e->h_in_ptrs[S + stream] = (const uint8_t *)d_chroma;
	} else {
		e->h_in_ptrs[stream] = e->d_in_y + (size_t)stream * e->cfg.input_luma_size;
		e->h_in_ptrs[S + stream] = e->d_in_uv + (size_t)stream * e->cfg.input_chroma_size;
	}
	for (bool &v : e->in_tab_valid)
		v = false;
	return 0;
}

extern "C" int cedar_enc_pipeline_depth(void)
{
	return CEDAR_ENC_PIPELINE_DEPTH;
}

extern "C" int cedar_enc_session_depth(const cedar_enc *e)
{
	if (!e || !e->configured)
		return -EINVAL;
	// gop_parallel: frame n needs frame n + (G-1)*K submitted before it can be delivered
	return e->mux ? e->mux->user_depth : e->depth;
}

// Cohort c of a big batch: the same kernels on the streams [S*c/C, S*(c+1)/C) -- every per-stream and
// per-unit array is one allocation with a fixed stride, so a cohort is an offset -- with scratch of

static long mux_frame_of(const GopMux &m, long q, int g) { return ((q / m.K) * m.G + g) * m.K + q % m.K; }
static long mux_step_of(const GopMux &m, long n) { return (n / m.K / m.G) * m.K + n % m.K; }

// submit lock-step steps while their frames are there and the session has room
static int mux_pump(cedar_enc *e)
{
	GopMux &m = *e->mux;
	while (e->fl_count < e->depth) {
		const long q = m.q_sub;
		// the step's last lane holds its latest frame; after a flush lanes without a frame idle
		const long first = mux_frame_of(m, q, 0);
		if (first >= m.n_sub)
			break;
		if (mux_frame_of(m, q, m.G - 1) >= m.n_sub && !m.flushed)
			break;
		for (int s = 0; s < m.S; s++)
			for (int g = 0; g < m.G; g++) {
				long n = mux_frame_of(m, q, g);
				const int lane = s * m.G + g;
				e->lane_gidx[lane] = n < m.n_sub ? e->ext.first_frame_index + n : -1;
				if (n >= m.n_sub)
					n = first; // an idle lane encodes lane 0's frame again; nobody reads the result
				const uint8_t *const *src = &m.in_ring[(size_t)(n % m.cap) * 2 * m.S];
				e->h_in_ptrs[lane] = src[s];
				e->h_in_ptrs[m.S * m.G + lane] = src[m.S + s];
			}
		for (bool &v : e->in_tab_valid)
			v = false;
		e->next_gop_start = q % m.K == 0;
		const int rc = submit_frame(e, nullptr);
		if (rc)
			return rc;
		m.q_sub++;
	}
	return 0;
}

// take the oldest step off the session and file its access units under their frames
static int mux_collect_step(cedar_enc *e)
{
	GopMux &m = *e->mux;
	const long q = m.q_col;
	const int rc = collect_frame(e);
	m.q_col++;
	for (int g = 0; g < m.G; g++) {
		const long n = mux_frame_of(m, q, g);
		if (n >= m.n_sub)
			continue; // idle lane
		int total = 0;
		for (int s = 0; s < m.S; s++) {
			std::vector<uint8_t> &o = m.out_ring[(size_t)(n % m.cap) * m.S + s];
			o.clear();
			if (rc >= 0) {
				const int lane = s * m.G + g;
				const uint8_t *src = e->h_out[e->res_host] + e->res_off[lane];
				o.assign(src, src + e->res_size[lane]);
				total += (int)e->res_size[lane];
			}
		}
		m.out_rc[(size_t)(n % m.cap)] = rc >= 0 ? total : rc;
	}
	return rc < 0 && rc != -EIO && rc != -ENOSPC && rc != -EOVERFLOW ? rc : 0; // a dropped step is reported per frame
}

static int mux_submit(cedar_enc *e)
{
	GopMux &m = *e->mux;
	if (m.flushed)
		return fail(e, -EINVAL, "submit after flush");
	if (m.n_sub - m.n_col >= m.cap)
		return fail(e, -EBUSY, "pipeline full: collect first");
	HIP_TRY(e, hipSetDevice(e->device));
	const size_t ring = (size_t)(m.n_sub % m.cap);
	const size_t ysz = (size_t)e->cfg.input_luma_size, csz = (size_t)e->cfg.input_chroma_size;
	bool any_host = false;
	hipStream_t s_in = e->lookahead ? e->s_in : e->s_main; // where every frame's k_input runs: copies land in front of it
	for (int s = 0; s < m.S; s++) {
		const uint8_t **dst = &m.in_ring[ring * 2 * m.S];
		if (m.user_in[s]) {
			dst[s] = m.user_in[s];
			dst[m.S + s] = m.user_in[m.S + s];
			continue;
		}
		if (!m.d_store_y) { // first frame that comes through the pinned buffers: a device frame per ring entry
			int rc2;
			if ((rc2 = dalloc(e, &m.d_store_y, (size_t)m.cap * m.S * ysz)) || (rc2 = dalloc(e, &m.d_store_uv, (size_t)m.cap * m.S * csz)))
				return rc2;
			// dalloc clears what it allocates on the NULL stream, and the session's streams do not wait for that one: the
			// copy of this very frame below overtook the clear now and then and frame 0 of every stream went into the
			// encoder partly zeroed (round 3's fuzz: 1 session in ~1000 with IDR look-ahead, 9 in 10 without; CONFIG's
			// own allocations are followed by a device-wide synchronisation)
			HIP_TRY(e, hipStreamSynchronize(nullptr));
		}
		uint8_t *dy = m.d_store_y + (ring * m.S + s) * ysz, *duv = m.d_store_uv + (ring * m.S + s) * csz;
		HIP_TRY(e, hipMemcpyAsync(dy, e->h_luma + (size_t)s * ysz, (size_t)e->cfg.src_width * e->cfg.src_height, hipMemcpyHostToDevice, s_in));
		HIP_TRY(e, hipMemcpyAsync(duv, e->h_chroma + (size_t)s * csz, (size_t)e->cfg.src_width * e->cfg.src_height / 2, hipMemcpyHostToDevice, s_in));
		dst[s] = dy;
		dst[m.S + s] = duv;
		any_host = true;
	}
	if (any_host) {
		HIP_TRY(e, hipEventRecord(e->ev_host_in, s_in));
		HIP_TRY(e, hipEventSynchronize(e->ev_host_in)); // the caller may refill the pinned buffers
	}
	m.out_rc[ring] = INT32_MIN;
	m.n_sub++;
	// queue what has become ready; when a step is ready and the session is full, wait for its oldest
	for (;;) {
		int rc = mux_pump(e);
		if (rc)
			return rc;
		const bool ready = mux_frame_of(m, m.q_sub, 0) < m.n_sub && mux_frame_of(m, m.q_sub, m.G - 1) < m.n_sub;
		if (!ready || e->fl_count < e->depth)
			return 0;
		if ((rc = mux_collect_step(e)))
			return rc;
	}
}

static int mux_collect(cedar_enc *e)
{
	GopMux &m = *e->mux;
	if (m.n_col >= m.n_sub)
		return fail(e, -EINVAL, "nothing in flight");
	const long n = m.n_col, q = mux_step_of(m, n);
	while (m.q_col <= q) {
		int rc = mux_pump(e);
		if (rc)
			return rc;
		if (m.q_sub == m.q_col)
			return fail(e, -EAGAIN, "frame %ld needs %ld more frame(s) submitted, or a flush", n,
				    mux_frame_of(m, m.q_col, m.G - 1) - m.n_sub + 1);
		if ((rc = mux_collect_step(e)))
			return rc;
	}
	m.n_col++;
	m.last = n;
	e->stats.frames = (uint64_t)m.n_col;
	const int rc = m.out_rc[(size_t)(n % m.cap)];
	if (rc < 0)
		return fail(e, rc, "frame %ld was dropped with its lock-step step", n);
	return rc;
}

static int do_submit(cedar_enc *e)
{
	if (!e->configured) // cedar.c:1039-1043
		return fail(e, -EINVAL, "CONFIG not run yet");
	if (e->mux)
		return mux_submit(e);
	return submit_frame(e, nullptr);
}

static int do_collect(cedar_enc *e)
{
	if (!e->configured)
		return -EINVAL;
	if (e->flusher && e->flusher->rc) { // a tail the flush thread could not enqueue: the frame it belongs to is lost
		const int rc = e->flusher->rc;
		e->flusher->rc = 0;
		return fail(e, rc, "the held-back part of a frame could not be enqueued");
	}
	return e->mux ? mux_collect(e) : collect_frame(e);
}

extern "C" int cedar_enc_submit(cedar_enc *e)
{
	if (!e)
		return -EINVAL;
	SessionLock lock(e);
	return do_submit(e);
}

extern "C" int cedar_enc_collect(cedar_enc *e)
{
	if (!e)
		return -EINVAL;
	SessionLock lock(e);
	return do_collect(e);
}

extern "C" int cedar_enc_set_frame_index(cedar_enc *e, int index)
{
	if (!e || !e->configured)
		return -EINVAL;
	SessionLock lock(e);
	if (e->mux)
		return fail(e, -EINVAL, "gop_parallel sessions number their frames themselves");
	if (index < 0 || index % e->cfg.keyframe_interval)
		return fail(e, -EINVAL, "frame %d is not the start of a GOP (keyframe interval %d)", index, e->cfg.keyframe_interval);
	// frames already in flight keep the index they were submitted under (InFlight::lane_gidx)
	e->ext.first_frame_index = index - e->frame_count;
	e->next_gop_start = true; // cedar.c:1047-1050: the next picture is an IDR picture (recorded with the frame: a replay keeps it)
	return 0;
}

extern "C" int cedar_enc_flush(cedar_enc *e)
{
	if (!e || !e->configured)
		return -EINVAL;
	SessionLock lock(e);
	if (e->mux)
		e->mux->flushed = true;
	hipSetDevice(e->device);
	return flush_deferred(e); // no further submit will come to put its first stripes in front of the held-back ones
}

extern "C" int cedar_enc_encode(cedar_enc *e)
{
	if (!e)
		return -EINVAL;
	SessionLock lock(e);
	if (e->configured && e->mux)
		return fail(e, -EINVAL, "gop_parallel sessions take frames through submit / collect / flush");
	if (e->fl_count != 0)
		return fail(e, -EBUSY, "frames in flight: collect them before a synchronous encode");
	// a failed encode leaves GOP counters and reference frames untouched
	// (cedar.c:1182-1191 return before :1193-1201): collect puts them back itself; a submit that
	// failed half way is undone here
	const GopState b{ e->frame_count, e->frame_p_count, e->cur, e->prev_slot, e->prev2_slot, e->idr_count };
	int r = do_submit(e);
	if (r < 0 && e->configured) {
		hipDeviceSynchronize();
		e->fl_count = 0;
		e->fl_head = 0;
		e->frame_count = b.frame_count;
		e->frame_p_count = b.frame_p_count;
		e->cur = b.cur;
		e->prev_slot = b.prev_slot;
		e->prev2_slot = b.prev2_slot;
		e->idr_count = b.idr_count;
		return r;
	}
	if (r >= 0)
		r = do_collect(e);
	return r;
}

extern "C" int cedar_enc_stream_output(cedar_enc *e, int stream, const uint8_t **data, int *size)
{
	if (!e || !e->configured || stream < 0 || stream >= e->ext.n_streams)
		return -EINVAL;
	if (e->mux) { // the access unit of the frame the last collect delivered
		if (stream >= e->mux->S || e->mux->last < 0)
			return -EINVAL;
		const std::vector<uint8_t> &o = e->mux->out_ring[(size_t)(e->mux->last % e->mux->cap) * e->mux->S + stream];
		if (data)
			*data = o.data();
		if (size)
			*size = (int)o.size();
		return 0;
	}
	if (data)
		*data = e->h_out[e->res_host] + e->res_off[stream];
	if (size)
		*size = (int)e->res_size[stream];
	return 0;
}

extern "C" void cedar_enc_close(cedar_enc *e)
{
	if (!e)
		return;
	if (e->configured) {
		hipSetDevice(e->device);
		{
			SessionLock lock(e);
			flush_deferred(e); // (frames abandoned in flight: the device must still be able to drain)
		}
		flusher_stop(e);
		hipDeviceSynchronize();
		if (getenv("CEDAR_DEBUG_LOG_DUMP"))
			dump_event_log(e);
	}
	pipe_token_release(e->device, e->holds_pipe_token, e->pipe_lock_fd);
	free_all(e);
	delete e;
}

extern "C" int cedar_enc_set_profiling(cedar_enc *e, int on)
{
	if (!e)
		return -EINVAL;
	// 0 off, 1 every stage, otherwise bit (1 + stage) selects a stage: events are stream markers,
	// a dozen of them per frame cost ~10 % of a pipelined 1080p stream
	// bits 8..15: time every n-th frame only (a marker behind a resident kernel costs its pipe ~20 us)
	e->profile_every = ((on >> 8) & 0xff) > 1 ? (on >> 8) & 0xff : 1;
	on &= 0xff;
	e->profiling = on != 0;
	e->profile_mask = on == 1 ? ~0u : (unsigned)on >> 1;
	return 0;
}

extern "C" int cedar_enc_get_stats(cedar_enc *e, struct cedar_enc_stats *out)
{
	if (!e || !out)
		return -EINVAL;
	*out = e->stats;
	out->wf_split = e->wf_split;
	out->deblock_planes_split = e->split_planes ? 1 : 0;
	out->deblock_rows = e->db_rows ? 1 : 0;
	out->intra_rows = e->ir_rows ? 1 : 0;
	out->src_direct = e->last_src_direct ? 1 : 0;
	out->entropy_fused = e->ent_fused ? 1 : 0;
	out->frames_pipelined = e->row_pipe ? 1 : 0;
	out->idr_lookahead = e->lookahead ? 1 : 0;
	out->depth = e->depth;
	out->deblock_streams = e->db_streams;
	return 0;
}

extern "C" int cedar_enc_reset_stats(cedar_enc *e)
{
	if (!e)
		return -EINVAL;
	memset(&e->stats, 0, sizeof(e->stats));
	return 0;
}

extern "C" int cedar_enc_debug_read(cedar_enc *e, int what, int stream, void *dst, int dst_size)
{
	if (!e || !e->configured || !dst || stream < 0 || stream >= e->ext.n_streams)
		return -EINVAL;
	if (e->fl_count)
		return fail(e, -EBUSY, "frames in flight");
	const Params &P = e->P;
	const void *src = nullptr;
	size_t n = 0;
	const int par = e->last_slot, rec = e->last_rec;
	switch (what) {
	case CEDAR_ENC_DBG_MBINFO: src = e->d_mbi[par] + (size_t)stream * P.n_mb; n = sizeof(MbInfo) * P.n_mb; break;
	case CEDAR_ENC_DBG_I4MODES: src = e->d_i4[par] + (size_t)stream * P.n_mb * 16; n = (size_t)16 * P.n_mb; break;
	case CEDAR_ENC_DBG_NNZ: src = e->d_nnz[par] + (size_t)stream * P.n_mb * NNZ_PER_MB; n = (size_t)NNZ_PER_MB * P.n_mb; break;
	case CEDAR_ENC_DBG_COEF: src = e->d_coef[par] + (size_t)stream * P.n_mb * COEF_PER_MB; n = (size_t)2 * COEF_PER_MB * P.n_mb; break;
	case CEDAR_ENC_DBG_RECON_Y: src = e->d_rec_y[rec] + (size_t)stream * P.plane_y_stride; n = (size_t)P.plane_y_stride; break;
	case CEDAR_ENC_DBG_RECON_UV: src = e->d_rec_uv[rec] + (size_t)stream * P.plane_uv_stride; n = (size_t)P.plane_uv_stride; break;
	case CEDAR_ENC_DBG_SRC_Y: src = e->d_src_y[par] + (size_t)stream * P.plane_y_stride; n = (size_t)P.plane_y_stride; break;
	case CEDAR_ENC_DBG_SRC_UV: src = e->d_src_uv[par] + (size_t)stream * P.plane_uv_stride; n = (size_t)P.plane_uv_stride; break;
	case CEDAR_ENC_DBG_MB_BITS: src = e->d_mb_bits[e->last_lane] + (size_t)stream * P.n_mb; n = sizeof(uint32_t) * P.n_mb; break;
	case CEDAR_ENC_DBG_MB_QP:
		if (!e->mb_qp)
			return fail(e, -ENOENT, "macroblocks of this session share the picture's QP");
		src = e->d_qp_eff[par] + (size_t)stream * P.n_mb; n = (size_t)P.n_mb; break;
	default: return -EINVAL;
	}
	if ((size_t)dst_size < n)
		return fail(e, -ENOSPC, "debug buffer needs %zu bytes", n);
	HIP_TRY(e, hipSetDevice(e->device));
	HIP_TRY(e, hipMemcpy(dst, src, n, hipMemcpyDeviceToHost));
	if (what == CEDAR_ENC_DBG_MBINFO) { // (an inter macroblock's chroma_mode says how its levels are stored: not part of the record)
		MbInfo *m = (MbInfo *)dst;
		for (int i = 0; i < P.n_mb; i++)
			if (m[i].type >= MB_P16x16)
				m[i].chroma_mode = 0;
	}
	if (what == CEDAR_ENC_DBG_COEF) {
		// k_inter only stores the levels of the block groups its coded block pattern names (the entropy kernels
		// read no others): what the buffer holds elsewhere is an earlier picture's. Shown as what it means: zero.
		std::vector<MbInfo> mi((size_t)P.n_mb);
		HIP_TRY(e, hipMemcpy(mi.data(), e->d_mbi[par] + (size_t)stream * P.n_mb, sizeof(MbInfo) * P.n_mb, hipMemcpyDeviceToHost));
		int16_t *cf = (int16_t *)dst;
		for (int i = 0; i < P.n_mb; i++, cf += COEF_PER_MB) {
			if (mi[i].type == MB_IPCM) { // (no level at all)
				memset(cf, 0, sizeof(int16_t) * COEF_PER_MB);
				continue;
			}
			if (mi[i].type < MB_P16x16)
				continue;
			if (mi[i].type == MB_P16x16 && (mi[i].chroma_mode & MB_LEVELS_INT8)) { // stored as bytes at the same indices: widen
				const int8_t *b8 = (const int8_t *)cf;
				for (int k = COEF_PER_MB - 1; k >= 0; k--)
					cf[k] = b8[k];
			}
			for (int g8 = 0; g8 < 4; g8++)
				if (!(mi[i].cbp & (1 << g8)))
					memset(cf + COEF_LUMA + 64 * g8, 0, 128);
			memset(cf + COEF_DC16, 0, 32);
			if (!(mi[i].cbp & 0x30))
				memset(cf + COEF_CDC, 0, 16);
			if (!(mi[i].cbp & 0x20))
				memset(cf + COEF_CAC, 0, 256);
		}
	}
	return (int)n;
}

extern "C" void cedar_enc_gen_pattern(int pattern, int w, int h, int frame, unsigned seed_offset,
				      uint8_t *luma, uint8_t *chroma)
{
	// 'A' static thumb-style picture (userspace/thumb.c:139-169 idea),
	// 'B' moving texture with global motion (3,2), 'C' LCG noise
	if (pattern == 'A') {
		for (int y = 0; y < h; y++)
			for (int x = 0; x < w; x++)
				luma[(size_t)y * w + x] = (x < 4 || y < 4 || x >= w - 4 || y >= h - 4) ? 0xEB : 0x10;
		for (int y = 0; y < h / 2; y++)
			for (int x = 0; x < w / 2; x++) {
				chroma[(size_t)y * w + 2 * x] = (uint8_t)x;
				chroma[(size_t)y * w + 2 * x + 1] = (uint8_t)y;
			}
	} else if (pattern == 'B') {
		uint32_t s = 0x5EED0000u + (uint32_t)frame + seed_offset;
		for (int y = 0; y < h; y++) {
			const int v = y + 2 * frame;
			uint8_t *row = luma + (size_t)y * w;
			for (int x = 0; x < w; x++) {
				const int u = x + 3 * frame;
				s = s * 1664525u + 1013904223u;
				row[x] = (uint8_t)(5 * u + 3 * v + 17 * ((u >> 4) ^ (v >> 4)) + (int)(s >> 29));
			}
		}
		for (int y = 0; y < h / 2; y++)
			for (int x = 0; x < w / 2; x++) {
				const int u = 2 * x + 3 * frame, v = 2 * y + 2 * frame;
				chroma[(size_t)y * w + 2 * x] = (uint8_t)(128 + ((u >> 1) & 63) - 32);
				chroma[(size_t)y * w + 2 * x + 1] = (uint8_t)(128 + ((v >> 1) & 63) - 32);
			}
	} else {
		uint32_t s = 0xC0FFEE00u + (uint32_t)frame + seed_offset;
		const size_t n = (size_t)w * h;
		for (size_t i = 0; i < n; i++) {
			s = s * 1664525u + 1013904223u;
			luma[i] = (uint8_t)(s >> 24);
		}
		for (size_t i = 0; i < n / 2; i++) {
			s = s * 1664525u + 1013904223u;
			chroma[i] = (uint8_t)(s >> 24);
		}
	}
}

