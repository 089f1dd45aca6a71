// cedarx_h264_encoder_amd/csrc/enc_session.h
// What the host-side files of the C-ABI share: the session object (struct cedar_enc), a frame in flight, the GOP lanes'
// rings, and the functions one file has for another.
//   enc_session.cpp   the entry points of include/cedar_h264enc.h, GOP lanes (gop_parallel), stats and debug reads
//   enc_config.cpp    CEDAR_IOCTL_CONFIG: parameter checks, buffers, streams and their roles, the schedule (cedar.c:732-866)
//   enc_headers.cpp   SPS / PPS / AUD bytes, what follows from a picture's QP, the rate law (cedar.c:187-223, :868-1030)
//   enc_submit.cpp    CEDAR_IOCTL_ENCODE as submit + collect: the launches of a frame, failure and replay (cedar.c:1032-1209)
#pragma once
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <vector>
#include <algorithm>

#include <sched.h>
#include <fcntl.h>
#include <sys/file.h>
#include <unistd.h>

#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include <hip/hip_runtime.h>

#include "../../include/cedar_h264enc.h"
#include "enc_internal.h"
#include "enc_plan.h"

using namespace cedar;

constexpr int MAX_COHORTS = 4; // see submit_frame
constexpr int DB_STREAMS = 3;
constexpr int AN_STREAMS = 3;
constexpr int ENT_LANES = 2;

inline int align_up(int x, int a) { return (x + a - 1) / a * a; }

// ---- put-bits: cedar_bytestream_write / _expgolomb / _expgolomb_signed ----
struct BitWriter {
	std::vector<uint8_t> buf;
	uint64_t nbits = 0;
	void put(uint32_t v, int n) // cedar.c:187-207
	{
		for (int i = n - 1; i >= 0; i--) {
			if ((nbits & 7) == 0)
				buf.push_back(0);
			if ((v >> i) & 1)
				buf.back() |= (uint8_t)(0x80 >> (nbits & 7));
			nbits++;
		}
	}
	void ue(uint32_t v) // cedar.c:209-215
	{
		v++;
		int len = 0;
		for (uint32_t t = v; t; t >>= 1)
			len++;
		put(v, 2 * len - 1);
	}
	void se(int32_t v) { ue(v > 0 ? (uint32_t)(2 * v - 1) : (uint32_t)(-2 * v)); } // cedar.c:217-223
	void trailing() // standard rbsp_trailing_bits; cedar.c:883-890 has a stray-byte quirk
	{
		put(1, 1);
		while (nbits & 7)
			put(0, 1);
	}
};

struct Timing {
	hipEvent_t ev[CEDAR_ENC_STAGE_COUNT][2];
	bool used[CEDAR_ENC_STAGE_COUNT];
};

// What cedar.c:1193-1201 advances after a frame that worked. A frame in flight keeps the values from
// before its submit: a failed collect puts them back (cedar.c:1182-1191 returns before the update).
struct GopState {
	int frame_count = 0, frame_p_count = 0, cur = 0, prev_slot = -1, prev2_slot = -1;
	unsigned idr_count = 0;
};

struct InFlight {
	bool active = false;
	bool idr = false;
	int host_out = 0;      // which host output buffer receives this frame
	std::vector<uint8_t> with_ps; // [n_streams] SPS+PPS precede this frame's slices
	GopState before;       // the session's GOP / reference state when this frame was submitted
	// where k_input found every stream of this frame ([2*n_streams]: luma, then chroma) and whether
	// that was the session's own device copy of the pinned input buffers: what a replay needs
	std::vector<const uint8_t *> in_used;
	std::vector<uint8_t> in_own;
	std::vector<long> lane_gidx; // [n_streams] global frame index per lock-step lane (-1: idle lane)
	bool gop_start = false;      // gop_parallel: every lane starts a GOP with this step, whatever the counter says
	int qp = 0;                  // the picture's QP
	int qp_bump = 0;             // ... of which this much because an earlier attempt had a macroblock above 3200 bits
	int lane = 0;          // entropy lane
	hipStream_t s_db = nullptr, s_ent = nullptr; // the streams that carry this frame's deblocking / entropy coding
	bool events = true;    // done_main / done_ent were recorded (frames not pipelined)
	int aud = 0;           // 0 none, 1 access unit delimiter of an IDR picture, 2 of a P picture
	hipEvent_t done_main = nullptr, done_ent = nullptr; // done_main: reconstruction deblocked (stream s_db)
	int ds = 0; // device slot (frame number % depth) holding this frame's device-side state
	SliceHdr *h_hdr = nullptr, *dv_hdr = nullptr;   // pinned + mapped per-frame slice headers, read by the kernels in place
	uint32_t *h_prefix = nullptr, *dv_prefix = nullptr;
	uint8_t *h_qp_s = nullptr, *dv_qp_s = nullptr;  // rate control over several streams: the picture's QP of every stream
	uint32_t *h_wf_err = nullptr; // pinned word the wavefront kernels write an error code to
	uint32_t *d_wf_err = nullptr; // the same word as the device sees it
	// completion flags in pinned host memory, written by 1-thread kernels at the end of the
	// frame's entropy ([0]) and deblock ([1]) work. collect polls them: hipEventSynchronize on a
	// stream that already holds later frames waits for those too.
	volatile uint32_t *h_done = nullptr;
	uint32_t *d_done = nullptr;
	uint32_t done_tag = 0;
	Timing tm;
};

// gop_parallel = G: the caller's S streams are encoded as S x G lock-step lanes, lane (s, g) taking
// every G-th closed GOP of stream s. Frames arrive and leave in stream order; in between they sit
// in a ring (device frames for the input, host bytes for the output). Lock-step step q encodes
// position t = q % K of the G GOPs of group q / K, i.e. user frame ((q / K) * G + g) * K + t on lane g.
struct GopMux {
	int G = 1, K = 1, S = 1;          // lanes per stream, GOP length, the caller's streams
	int cap = 0;                      // frames the rings hold
	int user_depth = 0;               // frames the caller may have in flight
	bool ahead = false;               // user_depth includes a second (G-1) K of look-ahead
	long n_sub = 0, n_col = 0;        // user frames submitted / collected
	long q_sub = 0, q_col = 0;        // lock-step steps submitted / collected
	bool flushed = false;
	std::vector<const uint8_t *> in_ring;        // [cap][2 * S] device frame of every stream: luma, chroma
	uint8_t *d_store_y = nullptr, *d_store_uv = nullptr; // [cap][S] device copies of frames that came in the pinned buffers
	std::vector<std::vector<uint8_t>> out_ring;  // [cap][S] access units
	std::vector<int> out_rc;                     // [cap] bytes of the frame (all streams) or -errno
	std::vector<const uint8_t *> user_in;        // [2 * S] what cedar_enc_set_device_input gave (nullptr: pinned buffers)
	long last = -1;                              // frame the last collect delivered
};

// The inputs of a frame that is encoded again after a frame before it failed (see recover()).
struct ReplayIn {
	std::vector<const uint8_t *> in;
	std::vector<uint8_t> own;
	std::vector<long> gidx;
	bool gop_start;
	int qp_bump = 0;
};

// Sessions that hold back the tail of a pipelined P picture (cedar_enc::DeferredTail) leave kernels on the
// device that wait for work only the NEXT submit enqueues. The caller only ever blocks in ENCODE
// (cedar.c:1178-1180), and between two calls it may do anything -- a device-wide synchronisation
// included, which would then wait for those kernels until their wall-clock bound. So the tail never stays
// held back for long: a thread of the session enqueues it `us` microseconds after the submit that held it
// back unless the next submit (or a collect / flush / close) has done so by then. Every entry point of such
// a session takes `mu`; sessions without a held-back tail have no thread and take no lock.
struct cedar_enc;
struct Flusher {
	std::mutex mu;
	std::condition_variable cv;
	std::thread th;
	bool stop = false, armed = false;
	std::chrono::steady_clock::time_point deadline;
	long us = 100;
	int rc = 0;           // what a flush from the thread returned, reported by the next collect
	uint64_t flushes = 0; // tails the thread enqueued (cedar_enc_stats would show a caller that pauses between submits)
};

struct cedar_enc {
	bool configured = false; // cedar.c:744-747
	int device = 0;
	bool device_set = false; // CONFIG got as far as hipSetDevice(device): only then is there anything of ours on a device
	struct cedar_enc_config cfg{};
	struct cedar_enc_config_ext ext{};
	Params P{};
	GopMux *mux = nullptr;   // gop_parallel sessions
	int S_user = 1;          // the caller's streams (ext.n_streams counts lock-step lanes)
	// GOP state (cedar.c:93-95)
	int frame_count = 0, frame_p_count = 0;
	// submits ever made: completion and progress tags derive from it, NOT from frame_count, which a
	// failed frame rolls back -- a retried frame must not find the tags of the failed attempt
	uint32_t submit_seq = 0;
	uint32_t slot_tag[CEDAR_ENC_PIPELINE_DEPTH] = {}; // progress tag of the frame that used a device slot last
	std::vector<long> lane_gidx; // [n_streams] global frame index of the next frame, per lane (mux sessions set it per step)
	bool next_gop_start = false; // gop_parallel: the next submit starts a GOP on every lane
	uint32_t fault_at = 0;       // tests (CEDAR_FAULT_INJECT_AT)
	int64_t rc_bits[64] = {};    // rate control: bits of the access unit of frame (index % 64) ...
	int rc_qp[64] = {};          // ... and the QP it was coded at
	bool rc_lanes = false;       // rate control of a gop_parallel session: one QP per group of G GOPs ...
	int64_t grp_sum[32] = {};    // ... per group (index = group % 32) the bits of its frames, each scaled to the configured QP,
	int grp_n[32] = {};          // how many frames that is,
	long grp_id1[32] = {};       // and which group it is, + 1 (0: none yet)
	bool rc_multi = false;       // rate control over several lock-step streams: the law runs per stream, the
	std::vector<int64_t> rc_bits_s; // kernels take the picture's QP per stream (Params::qp_s); [stream][64] as above
	std::vector<int> rc_qp_s;
	int64_t rc_target = 0;       // bits per frame (0: fixed QP)
	int cohorts = 1;             // big batches go through the stages in this many groups of streams (submit_frame)
	size_t stream_out_cap = 0;   // bytes of output per stream: cohort c packs its streams from (first stream) x this on
	int cur = 0; // reconstruction buffer being written (cedar.c:117-119)
	int depth = 4;  // frames in flight this session allows (<= CEDAR_ENC_PIPELINE_DEPTH) = device slots
	int nrec = 5;   // reconstruction buffers (depth + 1)
	int chroma_qp_offset = 4;
	// device memory
	uint8_t *d_in_y = nullptr, *d_in_uv = nullptr; // device copies of the pinned input buffers: [depth][n_streams]
	uint8_t **d_in_ptrs = nullptr; // [depth][2*n_streams]
	std::vector<const uint8_t *> h_in_ptrs;
	std::vector<const uint8_t *> in_tab[CEDAR_ENC_PIPELINE_DEPTH]; // per slot: what d_in_ptrs holds
	bool in_tab_valid[CEDAR_ENC_PIPELINE_DEPTH] = {};
	// per frame in flight (slot): source planes, records, levels, wavefront sync + progress words
	uint8_t *d_src_y[CEDAR_ENC_PIPELINE_DEPTH] = {}, *d_src_uv[CEDAR_ENC_PIPELINE_DEPTH] = {};
	// reconstructions rotate through depth + 1 buffers: the one a frame writes was last touched by
	// frames that have been collected (written by frame n - nrec, searched by frame n - nrec + 1)
	uint8_t *d_rec_y[CEDAR_ENC_PIPELINE_DEPTH + 1] = {}, *d_rec_uv[CEDAR_ENC_PIPELINE_DEPTH + 1] = {};
	MbInfo *d_mbi[CEDAR_ENC_PIPELINE_DEPTH] = {};
	// regions of interest
	int8_t *d_roi_off = nullptr;
	QpRow *d_qtab = nullptr;
	uint8_t *d_qp_eff[CEDAR_ENC_PIPELINE_DEPTH] = {}, *d_qp_t[CEDAR_ENC_PIPELINE_DEPTH] = {};
	bool mb_qp = false; // macroblocks differ in QP: regions of interest and / or MAD classes
	int8_t *d_i4[CEDAR_ENC_PIPELINE_DEPTH] = {};
	uint8_t *d_nnz[CEDAR_ENC_PIPELINE_DEPTH] = {};
	int16_t *d_coef[CEDAR_ENC_PIPELINE_DEPTH] = {};
	// entropy coding scratch, one set per entropy lane (frames alternate between the lanes' streams)
	uint32_t *d_slot[ENT_LANES] = {}, *d_mb_bits[ENT_LANES] = {};
	uint8_t *d_skip[ENT_LANES] = {};
	uint32_t *d_bit_off[ENT_LANES] = {}, *d_run[ENT_LANES] = {}, *d_unit_bits[ENT_LANES] = {};
	bool last_src_direct = false; // ... and the last frame was
	bool main_prio = false;      // CEDAR_STREAM_PRIO=1: the analysis stream at the greatest stream priority, roles in classes (setup_streams)
	bool tail_gate = false;      // a stripe's launch waits at its end for the next stripe's gate (emit_stripes; CEDAR_TAIL_GATE)
	bool db_first = false;        // frames back to back: the deblock kernel is enqueued before the entropy chain (CEDAR_DB_FIRST)
	bool src_direct_ok = false;  // the width is a multiple of 16: frames may be read where the caller put them (CEDAR_SRC_DIRECT=0: always through k_input)
	// k_pack_out's workgroups per unit: twice what the session's pictures of the kind (IDR, P) have needed so far -- a guess by
	// size before the first one is back; whatever a picture needs beyond that its last workgroup does alone (same bytes)
	int pk_seen[2] = { 0, 0 };
	bool ent_fused = true;       // k_pack_out behind k_scan (CEDAR_ENTROPY_FUSED=0: k_zero .. k_esc_write + k_copy_out through the RBSP buffer)
	uint64_t *d_ent_rec[ENT_LANES] = {}, *d_pk_agg[ENT_LANES] = {}, *d_pk_unit[ENT_LANES] = {};
	uint32_t *d_pk_first[ENT_LANES] = {};
	uint8_t *d_rbsp[ENT_LANES] = {};
	uint32_t *d_esc[ENT_LANES] = {}, *d_nal_off[ENT_LANES] = {}, *d_nal_size[ENT_LANES] = {}, *d_stream_size[ENT_LANES] = {};
	int ent_lanes = 1;
	hipStream_t s_ent2 = nullptr; // lane 1 (an alias of the spare stream s_db[2]); lane 0 is s_ent
	uint8_t *d_out[CEDAR_ENC_PIPELINE_DEPTH] = {}; // compact bytestreams, one per frame in flight
	uint8_t *d_pred4 = nullptr;
	uint32_t *d_abort_all = nullptr; // Params::abort_all
	uint32_t *d_dbg_log = nullptr;   // Params::dbg_log (CEDAR_DEBUG_HANG)
	uint32_t *d_wf_sync[CEDAR_ENC_PIPELINE_DEPTH] = {};
	uint64_t *d_wf_hbuf[CEDAR_ENC_PIPELINE_DEPTH] = {};
	uint64_t *d_db_hbuf[CEDAR_ENC_PIPELINE_DEPTH] = {}; // k_deblock_rows: tagged bottom rows, band to band
	uint64_t *d_ir_hbuf[CEDAR_ENC_PIPELINE_DEPTH] = {}; // k_intra_rows: tagged last rows, band to band, both planes
	uint8_t *d_dummy_rec = nullptr;  // 24 rows: what idle k_inter waves reconstruct into
	uint32_t *d_stripe_tab = nullptr; // [inter_bands][row_num + 1] running macroblock counts of the diagonal stripes
	std::vector<int> stripe_s;        // [inter_bands + 1] stripe boundaries (diagonals)
	std::vector<int> stripe_n;        // macroblocks per stripe and stream
	uint32_t *d_prog[CEDAR_ENC_PIPELINE_DEPTH] = {}; // [0] prog_inter, [16..) prog_db[2][n_units]
	bool row_pipe = false;      // consecutive frames overlap along the wavefront diagonals (few units)
	bool holds_pipe_token = false;
	int pipe_lock_fd = -1;      // the device's lock file while this session is its one pipelined session across processes
	int db_streams = 1;         // deblock streams that really run beside s_main and each other
	int inter_bands = 1;
	int prev_slot = -1, prev2_slot = -1; // slots of the two frames submitted last
	int wf_split = 1, wf_split_intra = 1;
	int db_pairs = 8;
	bool db_rows = true;   // deblocking with a wave per macroblock row (k_deblock_rows); CEDAR_DEBLOCK_ROWS=0: the band-pair kernel
	int db_bands = 1;      // ... its workgroups per unit and plane: the bands of 4 rows of the tallest slice
	int db_lds_kb = 16;    // ... KiB of LDS each claims
	bool ir_rows = true;   // intra reconstruction as a wavefront of 4x4 blocks (k_intra_rows); CEDAR_INTRA_ROWS=0: k_intra_recon2
	int ir_lds_kb = 16;
	int hb_slots = WF_MAX_SPLIT - 1; // workgroup boundaries per unit the hand-off buffers have room for
	// Pipelined P pictures: the last stripes of a picture's analysis (from defer_from on), its final gate
	// and its entropy launches are held back and enqueued BETWEEN the first stripes of the next picture:
	// a picture's last stripe is gated on the END of the previous picture's deblocking, and on an in-order
	// stream the next picture's first stripe -- which only needs that deblocking a quarter through -- would
	// sit behind it. (submit_frame; emitted at the latest when the frame is collected.)
	bool pcm_session = false;    // the session takes I_PCM macroblocks (picture_takes_pcm says in which pictures)
	uint32_t *d_cut_votes[CEDAR_ENC_PIPELINE_DEPTH] = {}; // scene cuts: the votes of the P picture in every device slot
	Flusher *flusher = nullptr;  // only sessions that hold tails back (defer_from < inter_bands)
	int defer_default = 0;       // stripes held back unless CEDAR_DEFER says otherwise (set with the stripe count)
	int defer_from = 0;          // first deferred stripe; inter_bands = nothing is deferred
	struct DeferredTail {
		bool active = false;
		Params Pc{};
		hipStream_t s_ana = nullptr, s_ent = nullptr;
		int slot = 0, ei = 0, next = 0;
	} deferred;
	// pinned host memory
	uint8_t *h_luma = nullptr, *h_chroma = nullptr;
	uint8_t *h_out[CEDAR_ENC_PIPELINE_DEPTH] = {}, *dv_out[CEDAR_ENC_PIPELINE_DEPTH] = {};     // pinned + mapped, and its device view
	uint32_t *h_sizes[CEDAR_ENC_PIPELINE_DEPTH] = {}, *dv_sizes[CEDAR_ENC_PIPELINE_DEPTH] = {};
	size_t out_cap = 0;
	// streams / events
	// s_main: input + analysis of all frames in order; s_db[frame % DB_STREAMS]: deblocking (the
	// deblock kernels of consecutive frames overlap); s_ent: entropy coding + read-back. The HIP
	// runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them
	// the null stream's); streams sharing a queue run back to back, so the session asks for 8
	// before the runtime starts and probes what it got (see config).
	hipStream_t s_main = nullptr, s_ent = nullptr, s_db[DB_STREAMS] = {};
	hipStream_t s_ana[AN_STREAMS] = {}; // analysis streams; s_ana[0] == s_main
	int an_streams = 1;
	// IDR pictures depend on no other frame: with look-ahead they get streams of their own (input
	// of every frame on s_in, an IDR picture's analysis on s_idr, its deblocking on s_db_idr) and run
	// beside the P pictures submitted before them
	bool lookahead = false;
	bool idr_deblock_inline = true;
	hipStream_t s_in = nullptr, s_idr = nullptr, s_db_idr = nullptr, s_idr2 = nullptr;
	// Cohorts (sessions whose frames run back to back, submit_frame): the deblocking and the entropy chain
	// of cohort c > 0 have streams of their own. Both are chains of dependent steps / short kernels that
	// leave most of the chip idle; on one stream the cohorts' chains queued behind each other and that
	// stream set the step time (32 x 1080p: deblock stream 82 %, entropy stream 87 % busy).
	hipStream_t s_db_c[MAX_COHORTS] = {}, s_ent_c[MAX_COHORTS] = {};
	bool batch_self_reset = false; // CEDAR_BATCH_SELF_RESET=1 (development): no memset / k_flag launches around the kernels of frames that run
	                               // back to back, the kernels reset and flag themselves as for pipelined frames. Measured 32 x 1080p: 166 M MB/s against 170 M.
	hipEvent_t ev_entropy[CEDAR_ENC_PIPELINE_DEPTH * MAX_COHORTS] = {};
	unsigned idr_count = 0;
	bool split_planes = true;
	hipEvent_t ev_host_in = nullptr; // pinned input buffers consumed
	hipEvent_t ev_base = nullptr;    // time zero of CEDAR_TRACE_TIMELINE
	// per device slot: source planes + sync words ready (s_main), records complete (s_main),
	// reconstruction deblocked (s_db)
	// (x MAX_COHORTS: index slot * MAX_COHORTS + cohort)
	hipEvent_t ev_input[CEDAR_ENC_PIPELINE_DEPTH * MAX_COHORTS] = {}, ev_analysis[CEDAR_ENC_PIPELINE_DEPTH * MAX_COHORTS] = {},
		   ev_deblocked[CEDAR_ENC_PIPELINE_DEPTH * MAX_COHORTS] = {};

	InFlight fl[CEDAR_ENC_PIPELINE_DEPTH];
	int fl_head = 0, fl_count = 0; // ring of frames in flight
	// results of the last collected frame
	std::vector<uint32_t> res_off, res_size;
	int res_host = 0;
	int last_slot = 0, last_rec = 0, last_lane = 0;
	std::vector<uint8_t> sps_pps;
	std::vector<uint8_t> aud_nal[2]; // [0] in front of an IDR picture, [1] of a P picture
	// stats
	bool profiling = false;
	unsigned profile_mask = ~0u;
	int profile_every = 1;
	cedar_enc_stats stats{};
	uint32_t last_total = 0;
	char err[256] = { 0 };
};

// every entry point of a session with a flush thread runs under its lock
struct SessionLock {
	std::unique_lock<std::mutex> lk;
	explicit SessionLock(cedar_enc *e)
	{
		if (e && e->flusher)
			lk = std::unique_lock<std::mutex>(e->flusher->mu);
	}
};

namespace cedar {

// enc_session.cpp
int fail(cedar_enc *e, int code, const char *fmt, ...);

#define HIP_TRY(e, call)                                                                  \
	do {                                                                              \
		hipError_t _r = (call);                                                   \
		if (_r != hipSuccess)                                                     \
			return fail((e), _r == hipErrorOutOfMemory ? -ENOMEM : -EIO, "%s: %s", #call, hipGetErrorString(_r)); \
	} while (0)

template <typename T>
int dalloc(cedar_enc *e, T **p, size_t count)
{
	hipError_t r = hipMalloc((void **)p, count * sizeof(T) + 64);
	if (r != hipSuccess)
		return fail(e, -ENOMEM, "hipMalloc(%zu): %s", count * sizeof(T), hipGetErrorString(r));
	hipMemset(*p, 0, count * sizeof(T) + 64);
	return 0;
}

// enc_headers.cpp
extern const uint16_t kQuantMF[6][3];
extern const uint8_t kDequantV[6][3];
extern const uint8_t kChromaQp[52], kAlpha[52], kBeta[52], kTc0[52][3];
int lambda_of_qp(int qp);
void append_nal(std::vector<uint8_t> &out, int ref_idc, int type, const std::vector<uint8_t> &rbsp);
void build_sps_pps(cedar_enc *e);
SliceHdr build_slice_header(const cedar_enc *e, bool idr, int frame_num, int idr_pic_id, int first_mb, int qp_delta);
void set_qp_params(Params &P, int qp, int chroma_qp_offset);
bool picture_takes_pcm(int qp, int chroma_qp_offset);
int64_t rc_pow2_6_q16(int d);
int rc_pick_qp(int qp0, int64_t target, const int64_t *bits, const int *qps, long n, int keyint);
int rc_group_lag(int keyint);
int rc_pick_qp_group(int qp0, int64_t target, const int64_t *sum, const int *cnt, const long *id1, long m, int keyint);

// enc_config.cpp
void pipe_token_release(int device, bool &holds, int &lock_fd);
void free_all(cedar_enc *e);
void fill_static_params(cedar_enc *e);
int config_impl(cedar_enc *e, struct cedar_enc_config *cfg, const struct cedar_enc_config_ext *ext_in);

// enc_submit.cpp
int submit_frame(cedar_enc *e, const ReplayIn *replay);
int collect_frame(cedar_enc *e);
int flush_deferred(cedar_enc *e);
void flusher_main(cedar_enc *e);
void flusher_stop(cedar_enc *e);
void dump_event_log(cedar_enc *e);

} // namespace cedar
