/*
 * oracle/h264dec.c -- stand-alone conformance decoder tool (baseline / CAVLC only).
 *
 *   h264dec <in.264> <out.nv12|-> [<source.nv12>]
 *
 * TEST INFRASTRUCTURE: the container has no third-party
 * H.264 decoder, and the reference's own README doubts its stream
 * (README.md:16-17), so this tool decodes what the encoder wrote with the
 * independently written decoder (h264o_dec.c), writes the cropped nv12 pictures
 * and, given the source, reports luma/chroma PSNR per picture and overall.
 * Exit code 0 = every access unit decoded; 1 = syntax error (message on stderr).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "h264o.h"

static double psnr(double sse, double n)
{
	return sse <= 0 ? 99.99 : 10.0 * log10(255.0 * 255.0 * n / sse);
}

int main(int argc, char **argv)
{
	FILE *fi, *fo = NULL, *fs = NULL;
	uint8_t *buf, *src = NULL;
	long size, pos = 0, au_start = 0;
	int pictures = 0;
	double sse_y = 0, sse_c = 0, n_y = 0, n_c = 0;
	h264o_dec *d;

	if (argc < 3) {
		fprintf(stderr, "Usage: %s <in.264> <out.nv12|-> [<source.nv12>]\n", argv[0]);
		return 2;
	}
	fi = fopen(argv[1], "rb");
	if (!fi) {
		perror(argv[1]);
		return 2;
	}
	if (strcmp(argv[2], "-"))
		fo = fopen(argv[2], "wb");
	if (argc > 3)
		fs = fopen(argv[3], "rb");
	fseek(fi, 0, SEEK_END);
	size = ftell(fi);
	fseek(fi, 0, SEEK_SET);
	buf = malloc((size_t)size + 4);
	if (!buf || fread(buf, 1, (size_t)size, fi) != (size_t)size)
		return 2;
	d = h264o_dec_open();
	/* an access unit ends after each slice NAL that completes a picture: feed NAL by
	 * NAL groups, cutting after every VCL NAL (types 1, 5) */
	while (pos + 4 <= size) {
		long next = pos + 4;
		int type;

		if (!(buf[pos] == 0 && buf[pos + 1] == 0 && buf[pos + 2] == 0 && buf[pos + 3] == 1)) {
			pos++;
			continue;
		}
		type = buf[pos + 4] & 0x1f;
		while (next + 4 <= size && !(buf[next] == 0 && buf[next + 1] == 0 && buf[next + 2] == 0 && buf[next + 3] == 1))
			next++;
		if (next + 4 > size)
			next = size;
		if (type == 1 || type == 5) {
			int r = h264o_dec_decode(d, buf + au_start, (int)(next - au_start));

			if (r < 0) {
				fprintf(stderr, "picture %d: %s\n", pictures, h264o_dec_error(d));
				return 1;
			}
			au_start = next;
			if (r == 1) {
				int cw = h264o_dec_width(d), w = h264o_dec_crop_width(d), h = h264o_dec_crop_height(d), y;
				const uint8_t *ly = h264o_dec_luma(d), *lc = h264o_dec_chroma(d);

				if (fo) {
					for (y = 0; y < h; y++)
						fwrite(ly + (size_t)y * cw, 1, (size_t)w, fo);
					for (y = 0; y < h / 2; y++)
						fwrite(lc + (size_t)y * cw, 1, (size_t)w, fo);
				}
				if (fs) {
					size_t fsz = (size_t)w * h * 3 / 2;
					double ey = 0, ec = 0;
					int x;

					if (!src)
						src = malloc(fsz);
					if (fread(src, 1, fsz, fs) == fsz) {
						for (y = 0; y < h; y++)
							for (x = 0; x < w; x++) {
								double e = (double)ly[(size_t)y * cw + x] - src[(size_t)y * w + x];
								ey += e * e;
							}
						for (y = 0; y < h / 2; y++)
							for (x = 0; x < w; x++) {
								double e = (double)lc[(size_t)y * cw + x] - src[(size_t)w * h + (size_t)y * w + x];
								ec += e * e;
							}
						printf("picture %4d  PSNR Y %.2f dB  C %.2f dB\n", pictures,
						       psnr(ey, (double)w * h), psnr(ec, (double)w * h / 2));
						sse_y += ey; sse_c += ec;
						n_y += (double)w * h; n_c += (double)w * h / 2;
					}
				}
				pictures++;
			}
		}
		pos = next;
	}
	printf("decoded %d pictures %dx%d", pictures, h264o_dec_crop_width(d), h264o_dec_crop_height(d));
	if (n_y > 0)
		printf(", PSNR Y %.2f dB C %.2f dB", psnr(sse_y, n_y), psnr(sse_c, n_c));
	{
		unsigned num, den;

		h264o_dec_frame_rate(d, &num, &den);
		if (den)
			printf(", %.3f frames/s (VUI %u/%u)", (double)num / den, num, den);
		if (h264o_dec_aud_count(d))
			printf(", %d access unit delimiters", h264o_dec_aud_count(d));
	}
	printf("\n");
	h264o_dec_close(d);
	free(buf);
	free(src);
	return 0;
}
