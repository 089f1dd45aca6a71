// cedarx_h264_encoder_amd/csrc/enc_plan.h
// The schedule policy of a session as PURE functions of its geometry, the device's CU count, what the stream
// probes found and the CEDAR_* development switches: how many frames in flight, how the two wavefront kernels
// are split over workgroups, which deblock form, whether frames are pipelined on the device, how many stripes
// and how many of them are held back, how many cohorts. No HIP, no session state: enc_config.cpp calls them
// from CONFIG, and tests/test_plan.py holds every threshold through cedar_enc_plan() on a box without a GPU
// (the 4K deadlock of round 2 -- more bands than wave pairs -- lived in exactly this code).
// Every number here was measured on MI355X; DESIGN.md 5.1-5.5 says with what.
#pragma once

#include "enc_internal.h"

namespace cedar {

constexpr int PLAN_UNSET_INIT = -2147483647 - 1;

struct PlanIn {
	int mb_w, mb_h;
	int slice_count, slice_first, slice_num;
	int n_streams;          // lock-step streams x GOP lanes
	int cus;                // compute units of the device
	int pinned_slot_bytes;  // bytes of pinned bytestream buffer one ring slot takes per stream
	// development switches (CEDAR_*): INT32_MIN = not set
	int env_row_pipe, env_pipe_depth, env_wf_split, env_wf_split_intra, env_db_pairs, env_deblock_streams,
	    env_deblock_planes, env_inter_bands, env_defer, env_cohorts, env_deblock_rows;
	int env_intra_rows = PLAN_UNSET_INIT; // CEDAR_INTRA_ROWS
	int mb_qp = 0;          // the session's macroblocks differ in QP (regions of interest, MAD classes, rate control over several streams)
};
constexpr int PLAN_UNSET = PLAN_UNSET_INIT;
constexpr int CEDAR_PLAN_MAX_DEPTH = 16; // = CEDAR_ENC_PIPELINE_DEPTH of the public header (asserted in enc_config.cpp)

struct Plan {
	int error;              // 0, or 1: a slice too tall for the deblock wavefront
	int n_units;
	int depth;              // frames in flight
	int wf_split, wf_split_intra, db_pairs;
	int split_planes;       // luma and chroma deblocking as separate workgroups
	int want_row_pipe;      // frames pipelined on the device, if the streams turn out to overlap
	int db_streams_wanted;  // deblock streams to probe for
	int defer_wanted;       // one unit: the tail of a P picture may be held back (wants a third deblock stream)
	int db_rows;            // deblocking with a wave per macroblock row (k_deblock_rows), a workgroup per band of 4 rows and plane
	int db_bands;           // ... bands of the tallest slice = its workgroups per unit and plane
	int db_lds_kb;          // ... KiB of LDS a workgroup claims: 96 = no second band on its CU, 16 = what its tiles take
	int ir_rows;            // intra reconstruction as a wavefront of 4x4 blocks, a wave per macroblock row (k_intra_rows)
	int ir_lds_kb;          // ... KiB of LDS a band claims (its bands are db_bands); 0: what it needs
	int idr_beside_ok;      // an IDR picture's deblocking may be resident beside its intra wavefront (a third picture of deblock workgroups)
};

struct PlanProbe {
	int db_streams_usable;  // deblock streams that really run beside the analysis, the entropy stream and each other
	int holds_pipe_token;   // this session is the one pipelined session of its device and process
};

struct PlanRun {
	int row_pipe;
	int db_streams;
	int inter_bands;        // stripes of a P picture's analysis
	int defer_default;      // stripes held back unless CEDAR_DEFER says otherwise
	int cohorts;
};

inline void plan_band_range(const PlanIn &in, int &nb_min, int &nb_max)
{
	nb_min = 1 << 30;
	nb_max = 0;
	for (int s = in.slice_first; s < in.slice_first + in.slice_num; s++) {
		int a, b;
		slice_rows(in.mb_h, in.slice_count, s, &a, &b);
		const int nb = (b - a + 3) / 4;
		nb_min = nb < nb_min ? nb : nb_min;
		nb_max = nb > nb_max ? nb : nb_max;
	}
}

// everything that is decided before the streams exist
inline Plan plan_static(const PlanIn &in)
{
	Plan p{};
	const int n_units = in.n_streams * in.slice_num;
	p.n_units = n_units;
	auto set = [](int v) { return v != PLAN_UNSET; };

	// Frames in flight. Sessions that pipeline their frames on the device (few units) take IDR pictures out of the
	// frame order -- they depend on nothing -- so an IDR picture submitted early enough costs no time; "early
	// enough" is its ~2.5 ms against ~0.2 ms per P picture (1080p), hence the deep ring. Everything else keeps 4.
	// Every slot owns a pinned bytestream buffer sized for the worst case: the ring's share of host memory stays
	// under 512 MB (4K: 33 MB a slot, eight 1080p streams: 67 MB).
	{
		const bool deep = set(in.env_row_pipe) ? in.env_row_pipe != 0 : n_units <= 8;
		int depth = set(in.env_pipe_depth) ? in.env_pipe_depth : deep ? CEDAR_PLAN_MAX_DEPTH : 4;
		const long long per_slot = (long long)in.pinned_slot_bytes * in.n_streams;
		if (!set(in.env_pipe_depth) && deep && (long long)depth * per_slot > (512ll << 20))
			depth = (int)((512ll << 20) / per_slot);
		if (depth < 4) depth = set(in.env_pipe_depth) ? depth : 4;
		if (depth < 2) depth = 2;
		if (depth > CEDAR_PLAN_MAX_DEPTH) depth = CEDAR_PLAN_MAX_DEPTH;
		p.depth = depth;
	}

	// Workgroups per unit for the wavefront kernels: one CU is instruction-bound by a single unit's wavefront, so
	// few units are spread over more CUs. Measured, 1080p (17 bands of 4 macroblock rows per unit): deblock with 8
	// wave pairs per workgroup: 1 unit 0.64 / 0.51 / 0.47 ms at 2 / 4 / 8 parts; 32 units 1.06 / 0.89 / 0.82 ms at
	// 1 / 2 / 4 parts. Up to 8 units the frames are pipelined, which wants the resident deblock workgroups of 3
	// frames within half of the CUs: 4 streams 55 M MB/s with 4 parts pipelined against 40 M with 8 parts back to
	// back; 8: 69 M (2 parts, pipelined) against 62-67 M; 16: pipelining loses (80 M against 91 M).
	int nb_min, nb_max;
	plan_band_range(in, nb_min, nb_max);
	auto clamp_split = [&](int split) {
		if (split > WF_MAX_SPLIT) split = WF_MAX_SPLIT;
		if (split > nb_min / 2) split = nb_min / 2;
		return split < 1 ? 1 : split;
	};
	p.wf_split = clamp_split(set(in.env_wf_split) ? in.env_wf_split
			: (n_units <= 2 ? 8 : n_units <= 5 ? 4 : n_units <= 8 ? 2 : n_units <= 64 ? 4 : n_units <= 128 ? 2 : 1));
	{ // intra reconstruction: every band a triple of its own where the split allows (a workgroup with more bands than
	  // triples serialises them: 3 triples, 1 unit: 7.1 ms at 4 parts against 3.3 ms at 6)
		int si = set(in.env_wf_split_intra) ? in.env_wf_split_intra : (nb_max + IR_TRIPLES - 1) / IR_TRIPLES;
		if (si > WF_MAX_SPLIT) si = WF_MAX_SPLIT;
		if (si > nb_min) si = nb_min;
		p.wf_split_intra = si < 1 ? 1 : si;
	}
	// Deblock workgroups: 8 wave pairs (a whole CU: one unit's wavefront as fast as it goes, what the pipelined
	// sessions of up to 8 units want) or 2 pairs -- a quarter of the LDS and of the waves, which finds room on a CU
	// beside other kernels' workgroups. Above 8 units frames run back to back and the small form wins: 16 x 1080p
	// 114 -> 120 M MB/s, 32: 134 -> 139, 64: 139 -> 146 (up to 8 units: 1 / 4 / 8 streams 36.0 / 82.6 / 99.9 M with
	// 8 pairs against 35.1 / 54.7 / 88.5 with 2).
	p.db_pairs = set(in.env_db_pairs) ? (in.env_db_pairs == 2 ? 2 : 8) : n_units > 8 ? 2 : 8;
	if (p.db_pairs < 8 && !set(in.env_wf_split) && (n_units > 8 || set(in.env_db_pairs))) {
		// small workgroups: as many as it takes to give every band a wave pair
		const int want = (nb_min + p.db_pairs - 1) / p.db_pairs;
		p.wf_split = want > WF_MAX_SPLIT ? WF_MAX_SPLIT : want < 1 ? 1 : want;
	}
	// A band runs at most DB2_XRING (+3) macroblocks ahead of the band below it (the hand-off ring in LDS), and a
	// workgroup with more bands than wave pairs starts band b + pairs when band b is through: on a row longer than
	// pairs x ring that is a cycle (3840 wide, 9 bands on 8 pairs: round 2's deadlock). Then every band needs a pair.
	auto db_safe = [&](int split, int pairs) { return (nb_max + split - 1) / split <= pairs || in.mb_w + 3 <= pairs * DB2_XRING; };
	auto widen = [&](int pairs) {
		while (!db_safe(p.wf_split, pairs) && p.wf_split < WF_MAX_SPLIT && p.wf_split < nb_min)
			p.wf_split++;
		return db_safe(p.wf_split, pairs);
	};
	if (!widen(p.db_pairs)) {
		p.db_pairs = 8;
		if (!widen(8))
			p.error = 1;
	}

	// Luma and chroma as separate workgroups (two concurrent wavefronts) while CUs are spare; both planes in every
	// wave pair from 32 units on, where the cohorts overlap deblocking with the other half's search: 32 x 1080p
	// 152 M MB/s against 136-139 M with one-plane workgroups, 16 (no cohorts): 122 with one plane against 112.
	p.split_planes = set(in.env_deblock_planes) ? in.env_deblock_planes != 0 : n_units < 32;

	// One stream, one slice: the last stripes of a P picture are held back and go out between the first ones of the
	// next (DeferredTail in enc_session.h), which wants a third deblock stream. 1080p, M MB/s: 5 stripes / 2
	// streams 36.0; 8 stripes none held, 3 streams 36.4; 8 stripes 3 held, 2 streams 37.6; 8 stripes 3 held, 3 streams
	// 42.1-43.6. CEDAR_DEFER=0 switches it off.
	// Which deblock kernel. k_deblock_rows gives every band of 4 rows a workgroup of 7 waves and every macroblock row a
	// wave: one picture's wavefront takes 0.21 ms instead of 0.40 (1080p), but it spends more instructions per
	// macroblock. Two kinds of session get it (1080p, M MB/s, same run each):
	// * frames PIPELINED, the workgroups of two pictures on 3/4 of the CUs, every band claiming enough LDS that no second
	//   band and no intra workgroup shares its CU (db_lds_kb 96; the analysis may): one or two 1080p streams (43 -> 50,
	//   62.5 -> 63.9), a 4K picture in 1 or 4 slices. Two pictures' deblocking are resident at a time and spin until
	//   their analysis comes, which is why not more: with 4, 5, 7 or 8 x 1080p pipelined (544 resident workgroups of 7 waves
	//   at 8 streams, several to a CU) the next picture's gate gives up after 3 s with every band of the picture in front
	//   between steps 32 and 48 and the later units' bands not yet started -- the cause is not established (waiting at
	//   wave priority 0 instead of 3 changes nothing; 3 streams run, at 80 M against 82 with the wave pairs).
	// * 6 to 23 units, frames BACK TO BACK (nothing resident waits for anything; LDS as the tiles need, 16 KB): 6 streams
	//   91 -> 104, 8: 107 -> 130, 12: 128 -> 156, 16: 143 -> 168 (6 frames in flight, below), 24: 163 = 163; one stream as
	//   8 GOP lanes 97 -> 119.
	//   Below 6 the wave pairs with pipelined frames win (3: 76 against 64, 4: 89 against 79).
	// From 24 units on the parallelism is across pictures and the step is the sum of k_inter, deblocking and entropy on
	// the same VALUs: 32 x 1080p 171.0 with it against 171.5 with k_deblock2_planes' fewer, fatter waves and cohorts.
	// CEDAR_DEBLOCK_ROWS: 0 never, 2 whatever the unit count (development).
	int nb_lo, nb_hi;
	plan_band_range(in, nb_lo, nb_hi);
	p.db_bands = nb_hi;
	// Which intra kernel. k_intra_rows steps in 4x4 blocks, a wave per macroblock row (enc_k_intra.hip): a 1080p I picture
	// is ~1150 block times instead of 254 macroblock times of ten blocks each. Sessions whose macroblocks differ in QP keep
	// k_intra_recon2 (the chain waves hold the quantiser of the picture in registers); so do pictures wider than 4480
	// samples (its row buffers must leave room for a deblock workgroup on the same CU, below). CEDAR_INTRA_ROWS=0: never.
	p.ir_rows = (set(in.env_intra_rows) ? in.env_intra_rows != 0 : 1) && !in.mb_qp && in.mb_w <= 280 && (long long)n_units * nb_hi <= WF_SYNC_WORDS - 16;
	p.ir_lds_kb = 0;
	// WHO MAY BE RESIDENT AND WAITING. A pipelined deblock kernel sits on its CUs and polls until its picture's analysis
	// comes; what it waits for -- the analysis of its picture, behind it the deblocking of the picture before, behind that
	// an IDR picture's intra wavefront -- must find room whatever is resident. The hardware deals the workgroups of a launch
	// round-robin over the 8 XCDs and each XCD places its share IN ORDER on its own 32 CUs: a launch stalls as soon as ONE
	// of its workgroups finds no room in ITS XCD, however empty the rest of the chip is. (Round 3's "cause not established"
	// hang of 4 to 8 pipelined 1080p streams, and the one session in ten of 3840x2160 that ended in -EIO after 3 s in round 4:
	// three pictures of 72 deblock workgroups claiming a CU each, then 72 intra workgroups claiming a CU each -- 288 on 256.
	// tools/dev/hang_hunt2.sh, the device's own log under CEDAR_DEBUG_HANG.) So, for sessions that pipeline on this kernel:
	//   * a band claims 81 KB of LDS: never two bands on a CU (a filter wave that shares its SIMD with another band's runs
	//     at a fraction of its speed), but always room for ONE intra workgroup (k_intra_rows: what it needs, <= 79 KB, one per
	//     CU by its registers), or two k_inter workgroups, beside it;
	//   * the deblock workgroups of the pictures that can be resident at a time -- one per deblock stream, and a third when an
	//     IDR picture's deblocking runs beside its intra wavefront -- are counted PER XCD, each launch rounded up, and must
	//     leave two CUs of every XCD alone; k_intra_recon2's workgroups (which share a CU with nobody) count as well.
	const int W_db = 2 * n_units * nb_hi; // deblock workgroups of a picture, both planes
	const int xcd_cus = in.cus / 8;
	auto px = [](int w) { return (w + 7) / 8; };
	const int old_intra = p.ir_rows ? 0 : px(n_units * p.wf_split_intra);
	const int wgs2 = 2 * W_db; // workgroups of two pictures
	const bool rows_fit = n_units <= 8 && wgs2 <= in.cus * 3 / 4 && 2 * px(W_db) + old_intra <= xcd_cus - 2;
	p.idr_beside_ok = rows_fit && 3 * px(W_db) + old_intra <= xcd_cus - 2;
	// (round 4, with the entropy chain at 3 launches: 4 streams 95.9 against 91.0 with the wave pairs pipelined, 5: 109.2 against 106.3,
	// 3: 78 against 82)
	const bool rows_b2b = !rows_fit && n_units >= 4 && n_units < 24;
	p.db_rows = (set(in.env_deblock_rows) ? in.env_deblock_rows == 2 || (in.env_deblock_rows != 0 && (rows_fit || rows_b2b)) : rows_fit || rows_b2b) &&
		    (long long)n_units * nb_hi <= WF_SYNC_WORDS - 16; // (a progress word per band in the sync region)
	p.db_lds_kb = rows_fit ? 81 : 16;
	// (9 to 23 units back to back on it: 6 frames in flight instead of 4 -- 16 x 1080p 160.3 -> 168.6 M MB/s, 8 and 12 no
	// more; 8 streams, which come out at 8 by the 512 MB rule above, gain nothing from 6, 8 or 12)
	if (p.db_rows && rows_b2b && n_units > 8 && !set(in.env_pipe_depth) && !set(in.env_row_pipe))
		p.depth = 6;
	p.defer_wanted = n_units == 1 && !(set(in.env_defer) && in.env_defer == 0);
	// (with k_deblock_rows two pictures' deblocking in flight are enough and nothing is held back: 1080p 47.3 M MB/s
	// with 2 streams, 46.2 with 3 and 3 of 8 stripes held back, 30.9 with 1)
	p.db_streams_wanted = set(in.env_deblock_streams) ? in.env_deblock_streams : p.db_rows ? 2 : p.defer_wanted ? 3 : 2;
	// Frame pipelining needs kernels of the analysis and of the deblock streams on the GPU at the same time, and it
	// must leave the GPU mostly free: a pipelined deblock kernel is resident (one CU per workgroup) while it waits
	// for its frame's analysis, so the deblock workgroups of all frames in flight may only take half of the CUs.
	// (k_deblock_rows: a band claims enough LDS that no second band shares its CU, the analysis may)
	p.want_row_pipe = set(in.env_row_pipe) ? in.env_row_pipe != 0
			  : p.db_rows	       ? rows_fit
					       : n_units <= 8 && 2 * n_units * p.wf_split * p.db_streams_wanted <= in.cus / 2;
	return p;
}

// ... and what follows once the streams have been probed
inline PlanRun plan_run(const PlanIn &in, const Plan &p, const PlanProbe &pr)
{
	PlanRun r{};
	auto set = [](int v) { return v != PLAN_UNSET; };
	int rows = 0; // macroblock rows of the slices this session codes
	for (int sl = in.slice_first; sl < in.slice_first + in.slice_num; sl++) {
		int a, b;
		slice_rows(in.mb_h, in.slice_count, sl, &a, &b);
		rows += b - a;
	}
	const int n_units = p.n_units, n_mbs = in.mb_w * rows;
	r.row_pipe = p.want_row_pipe && pr.holds_pipe_token && pr.db_streams_usable >= 1;
	r.db_streams = pr.db_streams_usable >= 1 && p.want_row_pipe && pr.holds_pipe_token ? (pr.db_streams_usable < p.db_streams_wanted ? pr.db_streams_usable : p.db_streams_wanted) : 1;
	// every stripe costs a gate and two dependent launches: small pictures take fewer (854x480 13.1 -> 13.5 M MB/s
	// with 3, 1080p equal from 3 to 8 while nothing is held back)
	// (k_deblock_rows: 3 at any size -- 1080p, 8 / 3 runs each, the driver's K 20 arguments and the default K 300: 5 stripes
	// 38.4 / 50.2 M MB/s, 4: 40.5 / 50.5, 3: 41.0 / 51.0, 2: 43.0 / 49.0, 1: 34.1 / 35.1; every stripe is two launches of
	// the 85 us a submit takes, and the K 20 region waits for its IDR picture's submit)
	// (round 4, k_pack_out and no k_input: 2 / 3 / 4 / 5 stripes 52.8 / 58.5 / 59.7 / 60.2 M at K 300, 50.1 / 56.3 / 55.4 / 55.8 at K 20)
	// (one unit only: 3840x2160 in 4 slices 90.4 M with 3 stripes, 83.5 with 5; in one slice 83.1 -> 87.7)
	int nb = set(in.env_inter_bands) ? in.env_inter_bands : (n_mbs >= 4000 && (n_units == 1 || !p.db_rows) ? 5 : 3);
	r.defer_default = 0;
	if (r.row_pipe && p.defer_wanted && r.db_streams >= 3 && n_mbs >= 4000) {
		if (!set(in.env_inter_bands))
			nb = 8;
		r.defer_default = nb >= 8 ? 3 : nb >= 6 ? 2 : 0;
	}
	if (nb > 64) nb = 64;
	r.inter_bands = r.row_pipe ? (nb < 1 ? 1 : nb) : 1;
	// Cohorts (sessions whose frames run back to back): two from 32 units on, with the small deblock form -- the
	// deblocking of one half then runs beside the search of the other: 32 x 1080p 1.87 -> 1.73 ms per step, 64:
	// 3.58 -> 3.18 ms. Not below: the deblock wavefront of a picture takes its time however few pictures there are,
	// and half of 16 streams' search is shorter than that (16: 120 -> 92 M MB/s). Three or four lose everywhere.
	int C = set(in.env_cohorts) ? in.env_cohorts : (n_units >= 32 && p.db_pairs == 2 ? 2 : 1);
	if (r.row_pipe || C < 1) C = 1;
	if (C > 4) C = 4;
	if (C > in.n_streams) C = in.n_streams;
	r.cohorts = C;
	return r;
}

} // namespace cedar
