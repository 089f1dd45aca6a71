/*
 * h264enc -- the reference's command line on the MI355X path.
 *
 *   ./h264enc <infile|-> <width> <height> <outfile>
 *
 * Same arguments, frame pump, messages and exit codes as userspace/h264enc.c
 * (lines 134-201); open()/ioctl()/mmap() on /dev/cedar_dev are replaced by the
 * C-ABI of include/cedar_h264enc.h. Two literals differ from ve_config()
 * (h264enc.c:61,66): profile 66 and CAVLC, which this path implements.
 * Optional knobs come from the environment so the four positional arguments
 * stay: H264ENC_QP, H264ENC_KEYINT, H264ENC_SLICES, H264ENC_SEARCH, H264ENC_DEVICE,
 * H264ENC_GOP_PARALLEL (G closed GOPs side by side, same bytes), H264ENC_FIRST_FRAME (global index
 * of the first frame when the stream is sharded by GOPs over several processes / GPUs).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "cedar_h264enc.h"

#define ALIGN(x, a) (((x) + ((a) - 1)) & ~((a) - 1))

static cedar_enc *cedar;
static struct cedar_enc_config config;

static int env_int(const char *name, int dflt)
{
	const char *v = getenv(name);

	return v && *v ? atoi(v) : dflt;
}

static int ve_config(int width, int height) /* h264enc.c:47-117 */
{
	struct cedar_enc_config_ext ext;
	int ret;

	memset(&config, 0, sizeof(config));
	memset(&ext, 0, sizeof(ext));
	config.src_width = width;
	config.src_height = height;
	config.src_format = CEDAR_ENC_FORMAT_NV12;
	config.dst_width = ALIGN(width, 16);
	config.dst_height = ALIGN(height, 16);
	config.profile = 66;
	config.level = (config.dst_width / 16) * (config.dst_height / 16) > 8192 ? 51 : 41;
	config.qp = env_int("H264ENC_QP", 24);
	config.keyframe_interval = env_int("H264ENC_KEYINT", 25);
	config.entropy_coding_mode = CEDAR_ENC_ENTROPY_CAVLC;
	ext.size = sizeof(ext);
	ext.device = env_int("H264ENC_DEVICE", 0);
	ext.slice_count = env_int("H264ENC_SLICES", 1);
	ext.search_range = env_int("H264ENC_SEARCH", 0);
	ext.repeat_headers = env_int("H264ENC_REPEAT_HEADERS", 0);
	ext.aud = env_int("H264ENC_AUD", 0);
	ext.fps_num = env_int("H264ENC_FPS_NUM", 0);
	ext.fps_den = env_int("H264ENC_FPS_DEN", ext.fps_num > 0 ? 1 : 0);
	ext.gop_parallel = env_int("H264ENC_GOP_PARALLEL", 0);
	ext.gop_low_latency = env_int("H264ENC_GOP_LOW_LATENCY", 0);
	ext.first_frame_index = env_int("H264ENC_FIRST_FRAME", 0);
	ext.strict_mb_bits = env_int("H264ENC_STRICT_MB_BITS", 0);
	ext.bitrate = env_int("H264ENC_BITRATE", 0);
	ext.intra_refresh = env_int("H264ENC_INTRA_REFRESH", 0);
	if (getenv("H264ENC_MAD")) { /* "t0,t1,t2,t3:d0,d1,d2,d3,d4": thresholds and the QP offset of each class */
		if (sscanf(getenv("H264ENC_MAD"), "%d,%d,%d,%d:%d,%d,%d,%d,%d", &ext.mad_th[0], &ext.mad_th[1], &ext.mad_th[2], &ext.mad_th[3],
			   &ext.mad_dqp[0], &ext.mad_dqp[1], &ext.mad_dqp[2], &ext.mad_dqp[3], &ext.mad_dqp[4]) != 9) {
			fprintf(stderr, "%s(): H264ENC_MAD wants \"t0,t1,t2,t3:d0,d1,d2,d3,d4\", got \"%s\"\n", __func__, getenv("H264ENC_MAD"));
			return -EINVAL;
		}
	}
	if (getenv("H264ENC_ROI")) { /* "x,y,w,h,dqp;..." in macroblocks, up to 4 regions */
		const char *r = getenv("H264ENC_ROI");

		while (r && *r) {
			int *v = ext.roi[ext.roi_count < 4 ? ext.roi_count : 0], n = 0;

			if (ext.roi_count >= 4 || sscanf(r, "%d,%d,%d,%d,%d%n", &v[0], &v[1], &v[2], &v[3], &v[4], &n) != 5 ||
			    (r[n] != ';' && r[n] != 0)) {
				fprintf(stderr, "%s(): H264ENC_ROI wants up to 4 regions \"x,y,w,h,dqp;...\", got \"%s\"\n", __func__,
					getenv("H264ENC_ROI"));
				return -EINVAL;
			}
			ext.roi_count++;
			r += n;
			if (*r == ';')
				r++;
		}
	}
	ext.subpel = env_int("H264ENC_SUBPEL", 0);
	ext.pcm = env_int("H264ENC_PCM", 0); /* -1: no I_PCM macroblocks */
	ext.scene_cut = env_int("H264ENC_SCENE_CUT", 0);

	ret = cedar_enc_config_ext(cedar, &config, &ext);
	if (ret) {
		fprintf(stderr, "%s(): CEDAR_IOCTL_CONFIG failed: %s (%s)\n", __func__, strerror(-ret),
			cedar_enc_last_error(cedar));
		return ret;
	}
	printf("Input Y: %dbytes -> %p\n", config.input_luma_size, (void *)config.input_luma);
	printf("Input C: %dbytes -> %p\n", config.input_chroma_size, (void *)config.input_chroma);
	printf("Bytestream: %dbytes -> %p\n", config.bytestream_size, (void *)config.bytestream);
	return 0;
}

static int read_frame(int fd, void *buffer, int size) /* h264enc.c:119-132 */
{
	int total = 0, len;

	while (total < size) {
		len = read(fd, (char *)buffer + total, size - total);
		if (len <= 0)
			return -1;
		total += len;
	}
	return total;
}

int main(int argc, char **argv)
{
	unsigned frame_count = 0;
	int width, height, fd_in, fd_out, luma_size, chroma_size, ret;

	if (argc != 5) {
		printf("Usage: %s <infile> <width> <height> <outfile>\n", argv[0]);
		return -1;
	}
	width = atoi(argv[2]);
	height = atoi(argv[3]);

	cedar = cedar_enc_open();
	if (!cedar) {
		fprintf(stderr, "%s(): failed to open the HIP encoder: %s\n", __func__, strerror(errno));
		return -1;
	}
	if (strcmp(argv[1], "-")) {
		fd_in = open(argv[1], O_RDONLY | O_LARGEFILE);
		if (fd_in == -1) {
			fprintf(stderr, "%s(): Failed to open input file %s: %s\n", __func__, argv[1], strerror(errno));
			return -1;
		}
	} else {
		fd_in = 0;
	}
	fd_out = open(argv[4], O_CREAT | O_RDWR | O_TRUNC, S_IRUSR | S_IWUSR | S_IRGRP | S_IROTH);
	if (fd_out == -1) {
		fprintf(stderr, "%s(): Failed to open output file %s\n", __func__, argv[4]);
		return -1;
	}
	ret = ve_config(width, height);
	if (ret)
		return ret;

	luma_size = width * height;
	chroma_size = luma_size / 2;
	/* The reference loop is strictly serial: read, ENCODE, write (h264enc.c:181-198).
	 * Here frame n+1 is read and uploaded while frame n is still being encoded:
	 * cedar_enc_submit() returns as soon as the input buffers may be refilled,
	 * cedar_enc_collect() delivers the oldest frame. Same bytes, same order.
	 * H264ENC_PIPELINE=0 restores the serial loop. */
	if (env_int("H264ENC_PIPELINE", 1) || env_int("H264ENC_GOP_PARALLEL", 0) > 1) {
		int in_flight = 0;
		const int depth = cedar_enc_session_depth(cedar);

		while (1) {
			int got = read_frame(fd_in, config.input_luma, luma_size) == luma_size &&
				  read_frame(fd_in, config.input_chroma, chroma_size) == chroma_size;

			if (got) {
				ret = cedar_enc_submit(cedar);
				if (ret < 0)
					fprintf(stderr, "%s(): %d: CEDAR_IOCTL_ENCODE failed: %s\n", __func__,
						frame_count + in_flight, strerror(-ret));
				else
					in_flight++;
			}
			if (!got)
				cedar_enc_flush(cedar);
			while (in_flight > (got ? depth - 1 : 0)) {
				const uint8_t *data;
				int size;

				ret = cedar_enc_collect(cedar);
				in_flight--;
				if (ret < 0) {
					/* h264enc.c:189-192: that frame is not written, the loop goes on; the session
					 * has dropped exactly it and codes the frames behind it again (cedar.c:1182-1201
					 * leaves the GOP counter and the reference swap untouched on failure) */
					fprintf(stderr, "%s(): %d: CEDAR_IOCTL_ENCODE failed: %s\n", __func__, frame_count, strerror(-ret));
					continue;
				}
				cedar_enc_stream_output(cedar, 0, &data, &size);
				printf("\rFrame %5d: %5dbytes", frame_count, ret);
				if (write(fd_out, data, size) != size)
					fprintf(stderr, "%s(): short write\n", __func__);
				frame_count++;
			}
			if (!got)
				break;
		}
		return 0;
	}
	while (1) {
		ret = read_frame(fd_in, config.input_luma, luma_size);
		if (ret != luma_size)
			break;
		ret = read_frame(fd_in, config.input_chroma, chroma_size);
		if (ret != chroma_size)
			break;
		ret = cedar_enc_encode(cedar);
		if (ret < 0) {
			fprintf(stderr, "%s(): %d: CEDAR_IOCTL_ENCODE failed: %s\n", __func__, frame_count, strerror(-ret));
		} else {
			printf("\rFrame %5d: %5dbytes", frame_count, ret);
			if (write(fd_out, config.bytestream, ret) != ret)
				fprintf(stderr, "%s(): short write\n", __func__);
			frame_count++;
		}
	}
	return 0;
}
