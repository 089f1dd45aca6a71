// cedarx_h264_encoder_amd/csrc/enc_internal.h
// Shared between the host session code (enc_session.h and the four .cpp files behind it) and the gfx950
// kernels (enc_kernels.hip). Not part of the C-ABI (include/cedar_h264enc.h).
#pragma once

#include <stdint.h>

namespace cedar {

enum : int { MB_I4x4 = 0, MB_I16x16 = 1, MB_P16x16 = 2, MB_PSKIP = 3,
	     MB_IPCM = -1 }; // intra like 0 and 1 ("type < MB_P16x16"), no refIdx like them

// Per-macroblock decision record (8 bytes). Written by the analysis kernels,
// read by deblock and entropy. Plays the role of the VE's MBINFO / MVBUF
// side buffers (kernel/cedar.c:640, :650-651; cedar_regs.h:57,73).
struct MbInfo {
	int8_t type;         // MB_*
	int8_t i16_mode;     // 0 V 1 H 2 DC 3 Plane
	int8_t chroma_mode;  // 0 DC 1 H 2 V 3 Plane
	int8_t cbp;          // bits 0-3 luma 8x8, bits 4-5 chroma
	int16_t mvx, mvy;    // quarter samples
};

// Quantised levels of one macroblock, int16, zig-zag order inside a block:
//   [0,256)   luma[16][16]   block index = coding order
//   [256,272) Intra16x16 DC
//   [272,280) chroma DC [2][4]
//   [280,408) chroma AC [2][4][16] (entry 0 of each block unused)
constexpr int COEF_LUMA = 0;
constexpr int COEF_DC16 = 256;
constexpr int COEF_CDC = 272;
constexpr int COEF_CAC = 280;
constexpr int COEF_PER_MB = 408;  // 816 bytes = 51 x 16 B
// Round 4: a P macroblock all of whose levels fit a byte (any picture that is not noise at a very low QP) stores them as
// int8 at the same INDICES -- the first 408 bytes of its 816, 26 chunks of 16 -- and says so in MbInfo::chroma_mode (which
// means nothing for an inter macroblock): half the bytes k_inter writes and the CAVLC kernels read.
constexpr int MB_LEVELS_INT8 = 0x40;

constexpr int NNZ_PER_MB = 24;    // 16 luma (raster bx+4*by), 4 Cb, 4 Cr
constexpr int MB_BITS_LIMIT = 3200; // H.264 A.3.1: bits of one macroblock_layer()
constexpr int PCM_HEADER_BITS = 100;   // I_PCM decision: what mb_type, modes / vector differences, cbp and mb_qp_delta can cost at most
constexpr int PCM_LIMIT_OPEN = 2900;   // ... in I pictures, where the count is taken on the open loop's levels
constexpr uint32_t PCM_MB_BITS_FLAG = 0x80000000u; // Params::mb_bits of an I_PCM macroblock: its header bits | this
constexpr int SLOT_BYTES = 2048;  // private CAVLC bit buffer of one macroblock
constexpr int SLOT_WORDS = SLOT_BYTES / 4;
constexpr int ESC_CHUNK = 32;     // bytes of RBSP per emulation-prevention thread
constexpr int PK_CHUNK = 2048;    // k_pack_out: bytes of RBSP a workgroup builds, escapes and writes out
constexpr int PK_BITS = PK_CHUNK * 8;
constexpr int MAX_UNITS = 1024;   // streams x slices per session
constexpr int WF_WAVES = 16;      // waves of one wavefront workgroup
constexpr int IR_TRIPLES = 2;     // intra reconstruction: (luma, chroma, mover) wave triples per workgroup, one per band of 4 rows
constexpr int DB2_XRING = 16;     // deblock: macroblocks of bottom rows a band keeps for the band below (power of two)
constexpr int WF_MAX_SPLIT = 16;  // workgroups sharing one unit's wavefront
constexpr int WF_SYNC_WORDS = 16 + MAX_UNITS * 32; // [0] ticket [1] abort [16..] progress: a word per workgroup, [unit][workgroups per unit]
constexpr int PROG_BANDS = 128;     // deblock bands (4 macroblock rows) per unit with a progress word
constexpr int PROG_ALL = 0xffff;     // progress value: everything through
constexpr int INTER_WAVES = 4;       // macroblocks (waves) per k_inter workgroup: a 1080p picture is workgroup-launch bound at 1
constexpr int IN_K_STREAMS = 8;     // up to this many streams pass their frame pointers as kernel arguments
constexpr int DBR_HB_GRANULES = 24;  // k_deblock_rows: 4 luma rows x 4 words + 2 chroma rows x 4 words per macroblock
constexpr int DBG_LOG_ENTRIES = 16384; // Params::dbg_log
constexpr int WF_HBUF_GRANULES = 12; // 8-byte granules per macroblock: 4x16 B luma + 2x16 B chroma

// Regions of interest (cedar_enc_config_ext.roi): macroblocks differ in QP, so nothing derived from
// the QP can sit in Params. One row per QP_Y, built once per session: the kernels look a macroblock's
// row up (and row[qpc] for its chroma). 64 bytes.
struct QpRow {
	int32_t mf[3], dv[3];       // quantiser / scaling by position class
	int32_t qbits, qshift;      // 15 + qp / 6, qp / 6
	int32_t lambda;
	int32_t qpc;                // QP_C of this QP_Y (the session's chroma_qp_index_offset applied)
	int32_t alpha, beta;        // deblocking thresholds at indexA = indexB = this QP
	int32_t tc0[3];             // ... and tc0 for bS 1..3
	int32_t pad;
};

// What one picture-slice header contributes in front of the macroblock data.
struct SliceHdr {
	uint64_t bits;   // right-aligned header bits (<= 64)
	int32_t nbits;
	int32_t pad;
};

// Everything a kernel needs, passed by value. Per-stream arrays are one
// allocation each with a fixed stride; stream s lives at base + s*stride.
struct Params {
	// geometry
	int32_t width, height;      // source
	int32_t cw, ch;             // coded (multiples of 16)
	int32_t mb_w, mb_h, n_mb;
	int32_t n_streams;
	int32_t slice_count;        // slices per picture
	int32_t slice_first, slice_num; // slices this session encodes
	int32_t row_first, row_num; // MB rows covered by those slices
	// policy
	int32_t qp, qpc, search_range, lambda;
	int32_t subpel;             // motion vector precision: 0 whole samples, 1 half, 2 quarter
	int32_t is_idr;
	uint32_t *cut_votes;        // scene cuts (cedar_enc_config_ext.scene_cut; nullptr: off): macroblocks of this P picture that their own mean predicts better than the best vector
	int32_t pcm;                // this picture takes I_PCM macroblocks (cedar_enc_config_ext.pcm, and its QP leaves their edges unfiltered)
	int32_t intra_refresh;      // N >= 3: macroblocks with (x + 2*y + refresh_t) % N == 0 of a P picture are coded intra; 0 off
	int32_t refresh_t;          // P pictures since the IDR (1 for the first)
	// quantiser, luma (index = position class 0 even/even, 1 odd/odd, 2 mixed)
	int32_t mf[3], dv[3], qbits, qshift;
	int32_t mfc[3], dvc[3], qbits_c, qshift_c;
	// deblocking thresholds at qp / qpc
	int32_t alpha_y, beta_y, tc0_y[3];
	int32_t alpha_c, beta_c, tc0_c[3];
	// device pointers
	const uint8_t *const *in_luma;   // [n_streams] tightly packed width x height; nullptr: use in_k
	const uint8_t *const *in_chroma; // [n_streams]
	const uint8_t *in_k[2 * IN_K_STREAMS]; // few streams: the frame pointers travel in the kernel arguments (luma, then chroma)
	uint8_t *src_y, *src_uv;         // padded source planes
	// src_direct: no padded planes and no k_input -- the width is a multiple of 16, so the caller's packed frame IS the
	// source plane of every column, and the rows below the picture are its last row: readers clamp the row to src_last_y /
	// src_last_c (the coded size's last row when the planes are padded: a no-op)
	int32_t src_direct, src_last_y, src_last_c;
	uint8_t *rec_y, *rec_uv;         // reconstruction being written
	const uint8_t *ref_y, *ref_uv;   // previous reconstruction (deblocked)
	MbInfo *mbi;
	// regions of interest (nullptr: every macroblock at the slice QP, all of the above holds)
	const int8_t *roi_off;           // [n_mb] QP offset of every macroblock of the picture (same for all streams)
	const uint8_t *qp_s;             // [n_streams] the picture's QP per stream (rate control over several streams); nullptr: qp
	int32_t row_rule;                // QPs differ INSIDE a picture (regions, MAD classes): the first macroblock of a row always carries mb_qp_delta
	const QpRow *qtab;               // [52]
	uint8_t *qp_t;                   // [n_streams][n_mb] QP_Y the macroblock is quantised with (k_intra_decide / k_inter)
	int32_t mad_on, mad_th[4], mad_dqp[5]; // MAD classes: thresholds on luma SAD / 256 against the chosen prediction, QP offset per class
	uint8_t *qp_eff;                 // [n_streams][n_mb] QP_Y the decoder holds for the macroblock (k_qp_eff)
	int8_t *i4modes;                 // [n_mb][16]
	uint8_t *nnz;                    // [n_mb][24]
	int16_t *coef;                   // [n_mb][COEF_PER_MB]
	uint32_t *slot;                  // [n_mb][SLOT_WORDS]
	uint32_t *mb_bits;               // [n_mb]
	uint8_t *skip;                   // [n_mb]
	int32_t *mvd;                    // [n_mb][2] quarter-pel differences
	// per-unit entropy state (unit = stream*slice_num + local slice)
	uint32_t *bit_off;               // [n_units][unit_mb_cap + 2]
	uint32_t *run_before;            // same shape
	uint32_t *unit_bits;             // [n_units] total RBSP bits
	uint8_t *rbsp;                   // [n_units][rbsp_cap]
	uint32_t *esc_cnt;               // [n_units][esc_chunks_cap]
	uint32_t *nal_off;               // [n_units] byte offset of the NAL in out
	uint32_t *nal_size;              // [n_units]
	uint32_t *stream_size;           // [n_streams + 3] bytes per stream; [n] total; [n+1] error flags; [n+2] units scanned (k_esc_scan) / chunks through (k_pack_out); [n+3] k_copy_out's workgroups / k_pack_out's tickets
	// the fused back end (k_pack_out; nullptr: k_zero .. k_esc_write + k_copy_out through the RBSP buffer)
	uint64_t *ent_rec;               // [n_units][unit_mb_cap + 2] per entry: bit position | (skip run, bits of the macroblock (0: skipped), I_PCM) << 32
	uint32_t *pk_first;              // [n_units][pk_chunks_cap] the entry that covers the first bit of every chunk of PK_CHUNK RBSP bytes
	uint64_t *pk_agg;                // [n_units][pk_chunks_cap] a chunk's escaping table | the frame's tag (db_tag) << 32
	uint64_t *pk_unit;               // [n_units] a unit's escape bytes | the frame's tag << 32
	int32_t pk_chunks_cap;           // chunks per unit the arrays hold
	int32_t pk_deliver;              // how k_pack_out's bytes reach the host: 0 plain stores, the flag from the next kernel of the stream; 1 write-through stores, waited for, the flag from its last workgroup
	int32_t pk_launch_chunks;        // workgroups per unit of this launch (a unit's last one takes the chunks beyond)
	uint8_t *out;                    // compact output: all streams back to back
	const SliceHdr *hdr;             // [n_units] (streams differ where they are GOP lanes: idr_pic_id)
	const uint32_t *prefix_len;      // [n_streams] bytes reserved in front (SPS+PPS)
	const uint8_t *pred4_tab;        // [9][16] intra 4x4 predictor table
	// wavefront kernels split over wf_split workgroups (CUs) per unit: ticket,
	// abort flag and per-workgroup progress words (zeroed before every launch),
	// and the write-through hand-off buffer for the rows crossing a workgroup boundary
	uint32_t *wf_sync;               // [3][WF_SYNC_WORDS]: region 0 intra_recon, 1 deblock (luma), 2 deblock chroma
	uint64_t *wf_hbuf;               // [n_units][wf_hb_slots][mb_w][WF_HBUF_GRANULES] deblock
	uint64_t *wf_hbuf_intra;         // same shape, intra reconstruction (the two kernels of an IDR picture overlap)
	int32_t wf_split;                // workgroups per unit: deblock
	int32_t wf_split_intra;          // workgroups per unit: intra recon
	int32_t db_pairs;                // filter/mover wave pairs per deblock workgroup: 8 (a whole CU) or 2
	int32_t wf_hb_slots;             // workgroup boundaries per unit the hand-off buffers have room for
	int32_t db_rows;                 // 1: deblocking with a wave per macroblock row (k_deblock_rows), a workgroup per band of 4 rows ...
	int32_t db_lds_kb;               // ... KiB of LDS each claims: what its tiles take, or enough that no second band shares its CU
	int32_t db_pub_mask;             // ... a band publishes its progress every (mask + 1) steps (enc_config.cpp: db_pub_step)
	int32_t db_bands;                // ... and plane, this many per unit (the bands of the session's tallest slice)
	uint64_t *db_hbuf;               // [n_units][db_bands][mb_w][DBR_HB_GRANULES] k_deblock_rows: bottom rows of a band for the band below,
	uint32_t db_tag;                 // ... every 8-byte granule = 4 bytes of samples | this frame's tag << 32 (never 0; the buffer starts zeroed)
	int32_t ir_rows;                 // 1: intra reconstruction as a wavefront of 4x4 blocks, a wave per macroblock row (k_intra_rows) ...
	int32_t ir_bands;                // ... a workgroup per band of 4 rows and plane: the bands of the session's tallest slice
	int32_t ir_lds_kb;               // ... KiB of LDS each claims (as db_lds_kb)
	uint64_t *ir_hbuf;               // [n_units][2 planes][ir_bands][4 * mb_w] k_intra_rows: the last row of a band for the band below, tagged with db_tag
	// Frame pipelining (nullptr = frames run back to back). The wavefront kernels sweep the
	// picture along diagonals (absolute macroblock coordinates), so progress is counted in
	// diagonals: "every macroblock with s < D is through" -- s = x + y for the deblock kernel and
	// the P picture's analysis between two of them (one macroblock of lag per row), s = x + 2*y
	// for the intra wavefront and the IDR picture's deblocking behind it. Words live in device
	// memory, are written with agent-scope release and polled by kernels of other streams.
	// A value is (frame tag << 16 | D), compared modulo 2^32: a slot's words never need zeroing
	// and the stale value of the frame that used the slot before reads as "nothing yet".
	uint32_t *prog_inter;            // [1] D of this frame's analysis + reconstruction (all streams)
	uint32_t *prog_db;               // [2][n_units][PROG_BANDS] D of every deblock band written back: luma, then chroma
	uint32_t *prog_intra;            // [n_units][PROG_BANDS] D of every intra-reconstruction band written back (IDR pictures)
	uint32_t *prog_intra_c;          // the same for the chroma planes where they are bands of their own (k_intra_rows); nullptr: prog_intra speaks for both
	const uint32_t *prev_db;         // the previous frame's prog_db (what k_gate waits for)
	uint32_t prog_tag, prev_tag;
	int32_t stripe_s0, stripe_s1;    // k_inter: this launch covers s0 <= x + y < s1 (stripe_s1 <= 0: the whole picture)
	const uint32_t *stripe_tab;      // [row_num + 1] macroblocks of the stripe in the rows before row r (per stream)
	// A stripe's launch doing the gate work of the NEXT stripe (cedar_enc::tail_gate): its first wave publishes that the
	// stripe before it is complete (gate_publish >= 0) and, its own macroblock done, waits for what the next stripe's gate
	// would have waited for (gate_next_need > 0) -- the launch ends no earlier than that gate would have passed
	int32_t gate_publish, gate_next_need;
	uint8_t *dummy_rec;              // 16 + 8 rows of cw bytes: where the idle waves of k_inter's last workgroup write
	int64_t dummy_mb;                // ... and the macroblock record they fill: one past the session's last stream
	uint32_t *wf_err;                // pinned host word of this frame: a hand-off that timed out leaves a non-zero code
	uint32_t *abort_all;             // device word of the session: set by the first wait that ran out, so that the waits of
	                                 // the frames queued behind give up at once instead of one time-out after the other
	// Frame pipelining keeps the deblock stream to ONE packet per frame (every packet queued behind a
	// resident kernel costs its hardware pipe ~20 us): the last workgroup of the deblock launch
	// zeroes the sync words of regions 1-2 for the slot's next use (db_self_reset) and stores
	// done_tag to done_flag (pinned host memory) instead of a memset before and a k_flag after
	uint32_t *done_flag;
	uint32_t done_tag;
	int32_t db_self_reset;
	int32_t fault_inject;            // tests only (CEDAR_FAULT_INJECT): 1 a deblock mover, 2 an intra mover stops reporting
	uint32_t *dbg_log;               // development (CEDAR_DEBUG_HANG): [0] entries written, [4 + 4 i ..] { code | a << 8, b, s_memrealtime }; nullptr: off
	// strides (elements of the respective type)
	int64_t plane_y_stride, plane_uv_stride; // bytes per stream
	int32_t unit_mb_cap;             // max macroblocks of one unit
	int32_t rbsp_cap;                // bytes per unit
	int32_t esc_chunks_cap;          // per unit
	int32_t out_cap;                 // bytes total
	int32_t nal_ref_idc, nal_type;
};

// slice s of n covers MB rows [r0, r1): the first (mb_h % n) slices get one
// extra row (3840x2160 in 4 slices: 34/34/34/33).
static inline
#ifdef __HIPCC__
__host__ __device__
#endif
void slice_rows(int mb_h, int n, int s, int *r0, int *r1)
{
	int base = mb_h / n, rem = mb_h % n;
	int start = s * base + (s < rem ? s : rem);
	*r0 = start;
	*r1 = start + base + (s < rem ? 1 : 0);
}

static inline
#ifdef __HIPCC__
__host__ __device__
#endif
int slice_of_row(int mb_h, int n, int row)
{
	int base = mb_h / n, rem = mb_h % n;
	int split = rem * (base + 1);
	return row < split ? row / (base + 1) : rem + (row - split) / base;
}

// launch wrappers implemented in enc_kernels.hip; all asynchronous on `stream`.
struct Launch {
	static int input(const Params &p, void *stream);
	static int intra_decide(const Params &p, void *stream);
	static int intra_recon(const Params &p, void *stream);
	static int inter(const Params &p, void *stream, int n_mbs); // n_mbs macroblocks: the picture, or the diagonals [stripe_s0, stripe_s1)
	static int qp_eff(const Params &p, void *stream);           // macroblocks differ in QP: QP_Y per macroblock as the decoder holds it, for mb_qp_delta
	// 1 wave: publish prog_inter = publish_d (if >= 0), then wait until every deblock band of the
	// previous frame has written back all macroblocks with x + y < need_d (if > 0)
	static int gate(const Params &p, void *stream, int publish_d, int need_d);
	// d_out[0 .. stream_size[n_streams]) and the size words -> pinned host memory, by the GPU itself
	// (coalesced 16-byte stores over PCIe): no hipMemcpyAsync on the per-frame path
	static int copy_out(const Params &p, void *stream, uint8_t *host_out, uint32_t *host_sizes);
	// 1 thread: *word = value with system scope (completion flags in pinned host memory)
	static int flag(void *stream, uint32_t *word, uint32_t value);
	// concurrency probe: does a kernel on `waiter` see a flag set by a later launch on `setter`?
	// (HIP maps streams onto few hardware queues; two streams on one queue run back to back)
	static int probe_overlap(void *waiter, void *setter, uint32_t *d_words, bool *overlaps);
	// us per launch of a chain of n tiny kernels on `chain` while `blocked` (may be null) holds a
	// running kernel with a packet behind it: queues that share a command-processor pipe show ~8x
	static int probe_pipe(void *blocked, void *chain, uint32_t *d_words, int n_chain, double *us_per_kernel);
	static int deblock(const Params &p, void *stream, int planes); // 1 luma, 2 chroma, 3 both
	static int entropy(const Params &p, void *stream); // cavlc, scan, pack, escape
	// ... and the bytes on the host: the fused back end (cavlc, scan, k_pack_out), or entropy() + copy_out()
	static int entropy_out(const Params &p, void *stream, uint8_t *host_out, uint32_t *host_sizes);
};

int launch_deblock_rows(const Params &p, void *stream); // enc_k_deblock.hip (Launch::deblock when Params::db_rows)
int launch_intra_rows(const Params &p, void *stream);   // enc_k_intra.hip (Launch::intra_recon when Params::ir_rows)

// host-built table for the 4x4 intra predictors: entry = kind<<4 | k
// kind 0 copy E[k], 1 (E[k]+E[k+1]+1)>>1, 2 (E[k-1]+2E[k]+E[k+1]+2)>>2, 3 DC
void build_pred4_table(uint8_t tab[9 * 16]);

} // namespace cedar
