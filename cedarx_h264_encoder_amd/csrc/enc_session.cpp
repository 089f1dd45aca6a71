// cedarx_h264_encoder_amd/csrc/enc_session.cpp
// Host side of the C-ABI (include/cedar_h264enc.h): what the reference does in
// its ioctl handlers, minus the register poking.
//   config validation + buffer manager   kernel/cedar.c:732-866, :605-704
//   header bit-writer (SPS/PPS/slice)    kernel/cedar.c:187-223, :868-1030
//   GOP counters, reference ping-pong    kernel/cedar.c:1047-1050, :1193-1201
//   encode = program + trigger + wait    kernel/cedar.c:1032-1209
// The "trigger + wait for IRQ" (cedar.c:1176-1180) becomes kernel launches on
// two HIP streams (analysis/deblock and entropy) and an event wait.
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

namespace {

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

constexpr int MAX_COHORTS = 4; // see submit_frame
constexpr int DB_STREAMS = 3;
constexpr int AN_STREAMS = 3;
constexpr int ENT_LANES = 2;

// ---- ITU-T H.264 constants needed on the host (clauses 8.5.9, 8.7.2) ----
const uint16_t kQuantMF[6][3] = { // class 0 even/even, 1 odd/odd, 2 mixed
	{ 13107, 5243, 8066 }, { 11916, 4660, 7490 }, { 10082, 4194, 6554 },
	{ 9362, 3647, 5825 },  { 8192, 3355, 5243 },  { 7282, 2893, 4559 },
};
const uint8_t kDequantV[6][3] = {
	{ 10, 16, 13 }, { 11, 18, 14 }, { 13, 20, 16 }, { 14, 23, 18 }, { 16, 25, 20 }, { 18, 29, 23 },
};
const uint8_t kChromaQp[52] = {
	0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23, 24, 25,
	26, 27, 28, 29, 29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39
};
const uint8_t kAlpha[52] = {
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 4, 4, 5, 6, 7, 8, 9, 10, 12, 13,
	15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144, 162, 182, 203, 226, 255, 255
};
const uint8_t kBeta[52] = {
	0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4,
	6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17, 18, 18
};
const uint8_t kTc0[52][3] = {
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

} // namespace

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

static void flusher_main(cedar_enc *e);
static void flusher_stop(cedar_enc *e);

namespace {

int fail(cedar_enc *e, int code, const char *fmt, ...)
{
	va_list ap;
	va_start(ap, fmt);
	vsnprintf(e->err, sizeof(e->err), fmt, ap);
	va_end(ap);
	return code;
}

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
	struct Rec { // depth-first over the 7! assignments
		const double *blocked_frac, *launches;
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
				if (taken[sidx])
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
	Rec{ blocked_frac, launches }.go(0, cur, taken, 0.0, slow, best, best_cost);
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

double now_ms()
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

} // namespace

namespace cedar {

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

static int config_impl(cedar_enc *e, struct cedar_enc_config *cfg, const struct cedar_enc_config_ext *ext_in);

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
	if (e->mb_qp) {
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
	}
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
	if ((rc = dalloc(e, &e->d_pred4, (size_t)9 * 16))) return rc;
	if ((rc = dalloc(e, &e->d_abort_all, (size_t)16))) return rc;
	if (getenv("CEDAR_DEBUG_HANG") && (rc = dalloc(e, &e->d_dbg_log, (size_t)4 + 4 * DBG_LOG_ENTRIES))) return rc;
	for (int i = 0; i < e->depth && ext.scene_cut > 0; i++)
		if ((rc = dalloc(e, &e->d_cut_votes[i], (size_t)4))) return rc;
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
	{
		e->prev_slot = e->prev2_slot = -1;
	}
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

// The session's streams and what the probes make of the plan: which deblock streams really run beside the analysis and the
// entropy stream (frame pipelining polls across them), which hardware queues share a pipe (the roles are dealt by that), the
// IDR look-ahead streams, cohorts.
static int setup_streams(cedar_enc *e, const struct cedar_enc_config *cfg, const PlanIn &pin, const Plan &plan)
{
	Params &P = e->P;
	(void)P;
	(void)cfg;
	HIP_TRY(e, hipStreamCreateWithFlags(&e->s_main, hipStreamNonBlocking));
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
		if (want && !e->holds_pipe_token) {
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
		want = want && e->holds_pipe_token;
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
		const PlanRun run = plan_run(pin, plan, PlanProbe{ usable, e->holds_pipe_token ? 1 : 0 });
		e->row_pipe = run.row_pipe != 0;
		e->db_streams = run.db_streams;
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
		e->inter_bands = run.inter_bands;
		e->defer_default = run.defer_default;
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
		}
		if (getenv("CEDAR_TRACE_SUBMIT"))
			fprintf(stderr, "config: row_pipe %d, deblock streams %d, analysis streams %d, stripes %d, look-ahead %d, depth %d, cohorts %d\n",
				(int)e->row_pipe, e->db_streams, e->an_streams, e->inter_bands, (int)e->lookahead, e->depth, e->cohorts);
	}
	return 0;
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
			const int want = envd ? atoi(envd) : e->defer_default;
			int kd = nb;
			if (e->row_pipe && e->cohorts == 1 && !e->mux && want > 0 && want < nb && nb >= 3 && one_front)
				kd = nb - want;
			// Safe only if the previous picture's deblocking can get as far as this picture's stripe j asks (k_gate:
			// diagonal stripe_s[j + 1] + 4, in every band that has such diagonals) on what is published of the
			// previous picture's analysis when that gate runs: its stripes up to kd + j - 1 are in front of the gate,
			// but the end of a stripe is published by the gate of the NEXT one, so diagonal stripe_s[kd + j - 1] it is.
			// How far ahead of the asked diagonal the analysis must be: the progress words move every 8 steps (+8), a
			// stager waits for the analysis of 5 steps ahead (+5), and -- k_deblock_rows -- every band is on the SAME
			// diagonal as the band above (one macroblock behind its last row, like the rows of a band). (k_deblock2_planes,
			// bands inside one workgroup, every band a diagonal behind: the margin it has always run with.)
			const int slack = e->db_rows ? 4 + 8 + 5 + 3 : bands_max + 14;
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

static int config_impl(cedar_enc *e, struct cedar_enc_config *cfg, const struct cedar_enc_config_ext *ext_in)
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
	HIP_TRY(e, hipDeviceSynchronize());

	// static device pointers
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

	e->rc_target = ext.bitrate > 0 ? (int64_t)ext.bitrate * (ext.fps_num > 0 ? ext.fps_den : 1) / (ext.fps_num > 0 ? ext.fps_num : 30) : 0;
	build_sps_pps(e);
	e->frame_count = 0; // cedar.c:815-816
	e->frame_p_count = 0;
	e->res_off.assign(S, 0);
	e->res_size.assign(S, 0);
	e->lane_gidx.assign(S, 0);
	e->S_user = S_user;
	if (G > 1) {
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
	}
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
// its own where a launch counts or compacts (wavefront tickets, size words, the packed output).
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

// The inputs of a frame that is encoded again after a frame before it failed (see recover()).
struct ReplayIn {
	std::vector<const uint8_t *> in;
	std::vector<uint8_t> own;
	std::vector<long> gidx;
	bool gop_start;
	int qp_bump = 0;
};

// The stripes [k0, k1) of a pipelined P picture's analysis: a gate (publish how far this picture's
// analysis is, wait for the previous picture's deblocking) and the search of the stripe, each.
static int emit_stripes(cedar_enc *e, Params &Pc, hipStream_t s_ana, int k0, int k1)
{
	const int S = e->ext.n_streams;
	for (int k = k0; k < k1; k++) {
		Pc.stripe_s0 = e->stripe_s[k];
		Pc.stripe_s1 = e->stripe_s[k + 1];
		Pc.stripe_tab = e->d_stripe_tab + (size_t)k * (Pc.row_num + 1);
		if (Launch::gate(Pc, s_ana, k > 0 ? Pc.stripe_s0 : -1, Pc.stripe_s1 + 4))
			return fail(e, -EIO, "k_gate launch failed");
		if (e->stripe_n[k] > 0 && Launch::inter(Pc, s_ana, e->stripe_n[k] * S))
			return fail(e, -EIO, "k_inter launch failed");
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
static int flush_deferred(cedar_enc *e)
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
static void flusher_main(cedar_enc *e)
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

static void flusher_stop(cedar_enc *e)
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

// every entry point of a session with a flush thread runs under its lock
struct SessionLock {
	std::unique_lock<std::mutex> lk;
	explicit SessionLock(cedar_enc *e)
	{
		if (e && e->flusher)
			lk = std::unique_lock<std::mutex>(e->flusher->mu);
	}
};

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

static int submit_frame(cedar_enc *e, const ReplayIn *replay)
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
	Params P = e->P;
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
	const bool pipe = e->row_pipe;
	if (pipe) {
		P.prog_inter = e->d_prog[ds];
		P.prog_db = e->d_prog[ds] + 16;
		P.prog_intra = e->d_prog[ds] + 16 + 2 * (size_t)S * e->ext.slice_num * PROG_BANDS;
		P.prog_intra_c = e->ir_rows ? e->d_prog[ds] + 16 + 3 * (size_t)S * e->ext.slice_num * PROG_BANDS : nullptr;
		P.prev_db = prev >= 0 ? e->d_prog[prev] + 16 : nullptr;
		P.prog_tag = seq << 16;
		P.prev_tag = prev >= 0 ? e->slot_tag[prev] : 0;
		e->slot_tag[ds] = P.prog_tag;
	}
	// Which streams carry this frame. With look-ahead an IDR picture has streams of its own and
	// waits for nothing but its input: it runs beside the P pictures submitted before it.
	const bool ahead = e->lookahead && idr;
	// (idr_deblock_inline: the IDR picture's deblocking follows its analysis on the same stream
	// instead of running beside it on a stream of its own, resident and polling)
	const bool db_late = ahead && e->idr_deblock_inline;
	hipStream_t s_idr = ahead && e->s_idr2 && (e->idr_count & 1u) ? e->s_idr2 : e->s_idr; // IDR pictures take turns
	hipStream_t s_db = db_late ? s_idr : ahead ? e->s_db_idr : e->s_db[e->frame_count % e->db_streams];
	hipStream_t s_ana = ahead ? s_idr : e->s_ana[e->frame_count % e->an_streams]; // frames take turns: each stream's gates wait for one frame only
	hipStream_t s_in = e->lookahead ? e->s_in : s_ana; // input of all frames, in order, behind no gate
	P.nal_ref_idc = idr ? 3 : 2; // cedar.c:987-990
	P.nal_type = idr ? 5 : 1;

	// ---- input: H2D of the pinned buffers unless the caller supplied device frames ----
	tm_begin(e, f, CEDAR_ENC_STAGE_INPUT, s_in);
	bool in_work = false, any_host = false; // (something of this frame was enqueued on s_in; ... from the pinned buffers)
	{
		const int rc_in = submit_input(e, f, P, replay, ds, s_in, in_work, any_host);
		if (rc_in)
			return rc_in;
	}
	// (k_input -- k_intra_decide without it -- also zeroes the ticket / abort / progress words of the intra wavefront)
	*f.h_wf_err = 0;
	P.wf_err = f.d_wf_err;
	P.hdr = f.dv_hdr; // a few words per frame: the kernels read them from pinned host memory
	P.prefix_len = f.dv_prefix;

	build_frame_headers(e, f, P, idr);
	// entropy lanes: frames alternate between two streams with scratch buffers of their own
	const int lane = e->ent_lanes > 1 && e->s_ent2 ? e->frame_count % e->ent_lanes : 0;
	hipStream_t s_ent = lane ? e->s_ent2 : e->s_ent;
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
	f.lane = lane;
	f.host_out = slot; // every frame in flight owns one host buffer until it is collected
	if (e->ext.scene_cut > 0)
		e->h_sizes[slot][S + 2] = 0; // (k_copy_out reports a P picture's votes there)
	f.s_ent = s_ent;
	f.s_db = s_db;
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
	const int C = e->cohorts;
	// frames back to back: sync words and size words are left zeroed, the completion flags raised, by the kernels
	// themselves as for pipelined frames (not with cohorts on streams of their own: the last cohort's kernel
	// would speak for kernels of other streams)
	const bool self_zero = !pipe && e->batch_self_reset;                      // the kernels leave their sync / size words zeroed
	const bool self_reset = self_zero && !e->s_db_c[1] && !e->s_ent_c[1];     // ... and raise the completion flags too
	// a held-back tail only waits for the first stripes of a pipelined P picture; anything else: out with it
	if (e->deferred.active && !(pipe && !idr && C == 1)) {
		const int rc_f = flush_deferred(e);
		if (rc_f)
			return rc_f;
	}
	for (int c = 0; c < C; c++) {
		bool tail_done = false;
		Params Pc = cohort_params(e, P, c);
		const int ei = ds * MAX_COHORTS + c, eprev = prev * MAX_COHORTS + c;
		// the streams of this cohort's deblocking and entropy chain (cedar_enc::s_db_c)
		hipStream_t s_dbc = !pipe && !ahead && e->s_db_c[c] ? e->s_db_c[c] : s_db;
		hipStream_t s_entc = !pipe && e->s_ent_c[c] ? e->s_ent_c[c] : s_ent;
		if (!Pc.src_direct && Launch::input(Pc, s_in))
			return fail(e, -EIO, "k_input launch failed");
		if (c == C - 1)
			tm_end(e, f, CEDAR_ENC_STAGE_INPUT, s_in);
		// (frames read where the caller put them and nothing copied: the input stream has nothing of this frame to wait for)
		if (!Pc.src_direct || in_work) {
			HIP_TRY(e, hipEventRecord(e->ev_input[ei], s_in));
			if (s_in != s_ana)
				HIP_TRY(e, hipStreamWaitEvent(s_ana, e->ev_input[ei], 0));
		}

		auto launch_deblock = [&]() -> int {
		// ---- in-loop deblocking: produces the next frame's reference ----
			// Row-pipelined it starts at once and polls the analysis progress itself; otherwise it
			// waits for the whole analysis.
			// No event here: a cross-stream wait resolves to a later point of s_main than the record.
			// The kernel needs nothing of this frame's s_main work up front: its sync words are zeroed
			// on its own stream, the progress words need no zeroing (frame tags).
			if (!pipe)
				HIP_TRY(e, hipStreamWaitEvent(s_dbc, e->ev_analysis[ei], 0));
			if (db_late) // same stream: everything is there when the kernel starts, nothing to poll
				Pc.prog_intra = Pc.prog_intra_c = nullptr;
			// Pipelined: ONE packet on the deblock stream per frame -- every packet queued behind a
			// resident kernel costs the hardware pipe ~20 us. The kernel's last workgroup zeroes the
			// sync words for the slot's next frame and raises the frame's done flag itself.
			// Frames back to back do the same (a memset or a 1-thread kernel that meets a chip full of k_inter
			// workgroups waits for a CU like everybody else: 32 x 1080p showed the wf_sync memset take 377 us
			// in front of a deblock kernel). Cohorts follow each other on one stream: the last one's flag says all.
			Pc.db_self_reset = pipe || self_zero ? 1 : 0;
			Pc.done_flag = pipe || (self_reset && c == C - 1) ? f.d_done + 1 : nullptr;
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
					if ((!ahead && e->s_db_c[c2] ? e->s_db_c[c2] : s_db) != s_dbc)
						HIP_TRY(e, hipStreamWaitEvent(s_dbc, e->ev_deblocked[ds * MAX_COHORTS + c2], 0));
				HIP_TRY(e, hipEventRecord(f.done_main, s_dbc));
				if (!self_reset && Launch::flag(s_dbc, f.d_done + 1, f.done_tag))
					return fail(e, -EIO, "k_flag launch failed");
			}
			return 0;
		};
		// Row-pipelined, the deblock kernel only polls progress words: it is queued BEFORE this
		// frame's analysis so that it is resident and waiting when the first stripe arrives.
		if (pipe && !db_late) {
			const int rc_db = launch_deblock();
			if (rc_db)
				return rc_db;
		}

		// ---- analysis + reconstruction ----
		// What this frame overwrites or reads of its predecessors: rec[cur] was last touched by frames
		// that have been collected (the buffers rotate through depth + 1); a P picture reads the
		// reconstruction frame n-1 is deblocking. Row-pipelined, a P picture follows that deblocking
		// band by band (k_gate); otherwise it waits for the whole frame. An IDR picture waits for nothing.
		if (prev >= 0 && !idr && !pipe)
			HIP_TRY(e, hipStreamWaitEvent(s_ana, e->ev_deblocked[eprev], 0));
		if (idr) {
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
		} else {
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
					dt.s_ent = s_ent;
					dt.slot = slot;
					dt.ei = ei;
					dt.next = kd;
					flusher_arm(e);
				} else if ((rc_s = emit_tail(e, Pc, s_ana, s_ent, slot, ei))) {
					return rc_s;
				}
				tail_done = true;
			} else if (Launch::inter(Pc, s_ana, Pc.mb_w * Pc.row_num * Pc.n_streams)) {
				return fail(e, -EIO, "k_inter launch failed");
			}
			if (c == C - 1 && !tail_done)
				tm_end(e, f, CEDAR_ENC_STAGE_INTER, s_ana);
		}
		if (tail_done) { // a pipelined P picture: emit_tail has (or will have) done the rest
			t_main = t_ent = now_ms();
			continue;
		}
		HIP_TRY(e, hipEventRecord(e->ev_analysis[ei], s_ana));
		t_main = now_ms();

		// (frames back to back: the deblocking is enqueued BEFORE the entropy chain -- see below)
		const bool db_first = (!pipe || db_late) && e->db_first;
		if (db_first) {
			const int rc_db = launch_deblock();
			if (rc_db)
				return rc_db;
		}
		// ---- entropy coding on its own stream, concurrent with deblocking ----
		HIP_TRY(e, hipStreamWaitEvent(s_entc, e->ev_analysis[ei], 0));
		if (c == 0)
			tm_begin(e, f, CEDAR_ENC_STAGE_ENTROPY, s_entc);
		// Pipelined: as few packets on the entropy stream as possible (see launch_deblock): the size
		// words are left zeroed and the done flag raised by k_copy_out's last workgroup
		if (!pipe && !self_zero)
			HIP_TRY(e, hipMemsetAsync(Pc.stream_size, 0, sizeof(uint32_t) * (Pc.n_streams + 4), s_entc));
		// macroblocks differ in QP: QP_Y as the decoder will hold it, for the mb_qp_delta of the next one
		if (Pc.roi_off && Launch::qp_eff(Pc, s_entc))
			return fail(e, -EIO, "k_qp_eff launch failed");
		// (frames back to back: every cohort's last kernel leaves its size words zeroed, the last one raises the flag)
		Pc.done_flag = pipe || (self_reset && c == C - 1) ? f.d_done : self_zero ? e->d_abort_all + 8 : nullptr;
		Pc.done_tag = f.done_tag;
		if (Launch::entropy_out(Pc, s_entc, e->dv_out[f.host_out] + (size_t)(S * c / C) * (C > 1 ? e->stream_out_cap : 0),
					e->dv_sizes[slot] + (size_t)c * (S + 2)))
			return fail(e, -EIO, "entropy launch failed");
		if (c == C - 1)
			tm_end(e, f, CEDAR_ENC_STAGE_ENTROPY, s_entc);
		if (!pipe && e->ev_entropy[ei])
			HIP_TRY(e, hipEventRecord(e->ev_entropy[ei], s_entc));
		if (!pipe && c == C - 1) {
			for (int c2 = 0; c2 < c; c2++)
				if (e->ev_entropy[ds * MAX_COHORTS + c2] && (e->s_ent_c[c2] ? e->s_ent_c[c2] : s_ent) != s_entc)
					HIP_TRY(e, hipStreamWaitEvent(s_entc, e->ev_entropy[ds * MAX_COHORTS + c2], 0));
			HIP_TRY(e, hipEventRecord(f.done_ent, s_entc));
			if (!self_reset && Launch::flag(s_entc, f.d_done, f.done_tag))
				return fail(e, -EIO, "k_flag launch failed");
		}
		t_ent = now_ms();

		if ((!pipe || db_late) && !db_first) {
			const int rc_db = launch_deblock();
			if (rc_db)
				return rc_db;
		}
	}

	f.active = true;
	f.idr = idr;
	e->fl_count++;
	e->last_slot = ds;
	e->last_rec = e->cur;
	e->last_lane = lane;
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
	if (e->d_dbg_log) { // the device's own account: code, frame tag, argument, microseconds since the first entry
		std::vector<uint32_t> lg((size_t)4 + 4 * DBG_LOG_ENTRIES);
		if (hipMemcpy(lg.data(), e->d_dbg_log, lg.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
			const size_t n = std::min<size_t>(lg[0], DBG_LOG_ENTRIES);
			std::vector<std::array<uint64_t, 3>> ev;
			for (size_t i = 0; i < n; i++)
				ev.push_back({ (uint64_t)lg[4 + 4 * i + 2] | (uint64_t)lg[4 + 4 * i + 3] << 32, lg[4 + 4 * i], lg[4 + 4 * i + 1] });
			std::sort(ev.begin(), ev.end());
			static const char *const names[12] = { "?", "input", "intra_decide", "intra_rows start", "intra_rows end", "deblock start", "deblock end", "gate start",
							       "gate end", "inter", "?", "?" };
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

static int collect_frame(cedar_enc *e)
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
	bool hip_failed = false;
	{
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
	}
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
	e->res_host = f.host_out;
	e->last_total = total;
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
	e->stats.frames++;
	e->stats.encode_wall_ms += now_ms() - t0;
	return (int)total; // cedar.c:1207-1208: bytes now valid at bytestream offset 0
}

// ---- gop_parallel: S streams x G closed GOPs side by side (see GopMux) ----

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
