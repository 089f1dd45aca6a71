/*
 * oracle/h264enc_scalar.c -- the scalar C macroblock loop behind the
 * reference's command line (BASELINE.json configs[0]: plumbing, no GPU, no
 * /dev/cedar). TEST INFRASTRUCTURE: it is the CPU twin of the product CLI
 * (cedarx_h264_encoder_amd/csrc/h264enc_cli.c) and is never shipped as a
 * fallback of it.
 *
 * Usage and behaviour follow userspace/h264enc.c:134-201:
 *   h264enc_scalar <infile|-> <width> <height> <outfile>
 * reads width*height luma bytes then width*height/2 chroma bytes per frame,
 * stops at the first short read, exit codes as the reference (-1 on usage or
 * open errors, 0 at EOF). Extra knobs come from the environment so the four
 * positional arguments stay: H264ENC_QP, H264ENC_KEYINT, H264ENC_SLICES,
 * H264ENC_SEARCH.
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "h264o.h"

static int read_frame(int fd, uint8_t *buffer, int size) /* h264enc.c:119-132 */
{
	int total = 0;

	while (total < size) {
		ssize_t len = read(fd, buffer + total, (size_t)(size - total));

		if (len <= 0)
			return -1;
		total += (int)len;
	}
	return total;
}

static int env_int(const char *name, int dflt)
{
	const char *v = getenv(name);

	return v && *v ? atoi(v) : dflt;
}

int main(int argc, char **argv)
{
	h264o_config cfg;
	h264o_enc *enc;
	uint8_t *luma, *chroma, *out;
	int width, height, fd_in, fd_out, err, cap, frame_count = 0;

	if (argc != 5) {
		printf("Usage: %s <infile> <width> <height> <outfile>\n", argv[0]);
		return -1;
	}
	width = atoi(argv[2]);
	height = atoi(argv[3]);
	if (strcmp(argv[1], "-")) {
		fd_in = open(argv[1], O_RDONLY | O_LARGEFILE);
		if (fd_in == -1) {
			fprintf(stderr, "%s(): Failed to open input file %s: %s\n", __func__, argv[1], strerror(errno));
			return -1;
		}
	} else {
		fd_in = 0;
	}
	fd_out = open(argv[4], O_CREAT | O_RDWR | O_TRUNC, 0644);
	if (fd_out == -1) {
		fprintf(stderr, "%s(): Failed to open output file %s\n", __func__, argv[4]);
		return -1;
	}
	h264o_default_config(&cfg, width, height);
	cfg.qp = env_int("H264ENC_QP", cfg.qp);
	cfg.keyint = env_int("H264ENC_KEYINT", cfg.keyint);
	cfg.slice_count = env_int("H264ENC_SLICES", cfg.slice_count);
	cfg.search_range = env_int("H264ENC_SEARCH", cfg.search_range);
	cfg.repeat_headers = env_int("H264ENC_REPEAT_HEADERS", 0);
	cfg.aud = env_int("H264ENC_AUD", 0);
	cfg.fps_num = env_int("H264ENC_FPS_NUM", 0);
	cfg.fps_den = env_int("H264ENC_FPS_DEN", cfg.fps_num > 0 ? 1 : 0);
	cfg.subpel = env_int("H264ENC_SUBPEL", 0);
	cfg.pcm = env_int("H264ENC_PCM", 0); /* -1: no I_PCM macroblocks */
	cfg.scene_cut = env_int("H264ENC_SCENE_CUT", 0);
	cfg.rc_lanes = env_int("H264ENC_GOP_PARALLEL", 0); /* (only the rate control law knows about GOP lanes: the bytes of a fixed-QP stream do not) */
	cfg.bitrate = env_int("H264ENC_BITRATE", 0);
	cfg.strict_mb_bits = env_int("H264ENC_STRICT_MB_BITS", 0);
	cfg.intra_refresh = env_int("H264ENC_INTRA_REFRESH", 0);
	if (getenv("H264ENC_MAD")) { /* "t0,t1,t2,t3:d0,d1,d2,d3,d4": thresholds and the QP offset of each class */
		if (sscanf(getenv("H264ENC_MAD"), "%d,%d,%d,%d:%d,%d,%d,%d,%d", &cfg.mad_th[0], &cfg.mad_th[1], &cfg.mad_th[2], &cfg.mad_th[3],
			   &cfg.mad_dqp[0], &cfg.mad_dqp[1], &cfg.mad_dqp[2], &cfg.mad_dqp[3], &cfg.mad_dqp[4]) != 9) {
			fprintf(stderr, "%s(): H264ENC_MAD wants \"t0,t1,t2,t3:d0,d1,d2,d3,d4\", got \"%s\"\n", __func__, getenv("H264ENC_MAD"));
			return -1;
		}
	}
	if (getenv("H264ENC_ROI")) { /* "x,y,w,h,dqp;..." in macroblocks, up to 4 regions */
		const char *r = getenv("H264ENC_ROI");

		while (r && *r) {
			int *v = cfg.roi[cfg.roi_count < 4 ? cfg.roi_count : 0], n = 0;

			if (cfg.roi_count >= 4 || sscanf(r, "%d,%d,%d,%d,%d%n", &v[0], &v[1], &v[2], &v[3], &v[4], &n) != 5 ||
			    (r[n] != ';' && r[n] != 0)) {
				fprintf(stderr, "%s(): H264ENC_ROI wants up to 4 regions \"x,y,w,h,dqp;...\", got \"%s\"\n", __func__,
					getenv("H264ENC_ROI"));
				return -1;
			}
			cfg.roi_count++;
			r += n;
			if (*r == ';')
				r++;
		}
	}
	enc = h264o_open(&cfg, &err);
	if (!enc) {
		fprintf(stderr, "%s(): config failed: %s\n", __func__, strerror(-err));
		return err;
	}
	cap = h264o_max_frame_bytes(enc);
	luma = malloc((size_t)width * height);
	chroma = malloc((size_t)width * height / 2);
	out = malloc((size_t)cap);
	if (!luma || !chroma || !out)
		return -ENOMEM;
	while (1) {
		int ret;

		if (read_frame(fd_in, luma, width * height) != width * height)
			break;
		if (read_frame(fd_in, chroma, width * height / 2) != width * height / 2)
			break;
		ret = h264o_encode(enc, luma, chroma, out, cap);
		if (ret < 0) {
			fprintf(stderr, "%s(): %d: encode failed: %s\n", __func__, frame_count, strerror(-ret));
		} else {
			printf("\rFrame %5d: %5dbytes", frame_count, ret);
			if (write(fd_out, out, (size_t)ret) != ret)
				fprintf(stderr, "%s(): short write\n", __func__);
			frame_count++;
		}
	}
	h264o_close(enc);
	free(luma);
	free(chroma);
	free(out);
	return 0;
}
