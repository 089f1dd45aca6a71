// tools/dev/sad_bench.hip -- issue rates of the byte-SAD instructions and of the LDS read
// shapes k_inter's motion search is built from, on gfx950 (development probe, not product code).
//   hipcc --offload-arch=gfx950 -O3 tools/dev/sad_bench.hip -o tools/dev/sad_bench && tools/dev/sad_bench
// Prints SIMD cycles per wave-instruction at 1, 2 and 4 waves per SIMD (s_memtime around a loop
// of N instructions on independent accumulators; the result is stored so nothing is elided).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define N_ITER 256

template <int KIND>
__global__ void k_bench(uint32_t *out, uint64_t *cyc, uint32_t seed)
{
	__shared__ __attribute__((aligned(16))) uint8_t lds[64 * 48 + 256];
	for (int i = threadIdx.x; i < (int)sizeof(lds) / 4; i += blockDim.x)
		((uint32_t *)lds)[i] = seed * 2654435761u + i * 40503u;
	__syncthreads();
	const int lane = threadIdx.x & 63;
	uint64_t a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3;
	uint32_t b0 = seed, b1 = seed * 3, b2 = seed * 5, b3 = seed * 7;
	uint32_t c0 = lane * seed, c1 = c0 + 77, c2 = c0 * 3, c3 = c0 ^ 0x5a5a5a5a;
	const uint8_t *base = lds + (lane % 9) * 4 + (lane / 9) * 48;
	const uint64_t t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
	for (int it = 0; it < N_ITER; it++) {
		if (KIND == 0) { // v_qsad_pk_u16_u8 x4
			a0 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)c1 << 32) | c0, b0, a0);
			a1 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)c2 << 32) | c1, b1, a1);
			a2 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)c3 << 32) | c2, b2, a2);
			a3 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)c0 << 32) | c3, b3, a3);
		} else if (KIND == 1) { // v_sad_u8 x4
			b0 = __builtin_amdgcn_sad_u8(c0, c1, b0);
			b1 = __builtin_amdgcn_sad_u8(c1, c2, b1);
			b2 = __builtin_amdgcn_sad_u8(c2, c3, b2);
			b3 = __builtin_amdgcn_sad_u8(c3, c0, b3);
		} else if (KIND == 2) { // v_alignbyte_b32 x4
			b0 = __builtin_amdgcn_alignbyte(c0, b0, 1);
			b1 = __builtin_amdgcn_alignbyte(c1, b1, 2);
			b2 = __builtin_amdgcn_alignbyte(c2, b2, 3);
			b3 = __builtin_amdgcn_alignbyte(c3, b3, 1);
		} else if (KIND == 3) { // v_add_u32 x4 (reference: a full-rate op)
			b0 += c0; b1 += c1; b2 += c2; b3 += c3;
		} else if (KIND == 4) { // 5 x ds_read_b32 (k_inter's row fetch) feeding one qsad group
			const uint32_t *rp = (const uint32_t *)(base + (it & 15) * 48);
			uint32_t w0 = rp[0], w1 = rp[1], w2 = rp[2], w3 = rp[3], w4 = rp[4];
			b0 += w0; b1 += w1; b2 += w2; b3 += w3 + w4;
		} else if (KIND == 5) { // 1 x ds_read_b128 + 1 x ds_read_b32 (16-byte aligned rows)
			const uint8_t *q = lds + (lane & 3) * 16 + (lane >> 2) * 48 + (it & 15) * 48;
			const uint4 v = *(const uint4 *)q;
			const uint32_t w4 = *(const uint32_t *)(q + 16);
			b0 += v.x; b1 += v.y; b2 += v.z; b3 += v.w + w4;
		} else if (KIND == 6) { // the k_inter inner row: 5 reads + 4 qsad
			const uint32_t *rp = (const uint32_t *)(base + (it & 15) * 48);
			uint32_t w0 = rp[0], w1 = rp[1], w2 = rp[2], w3 = rp[3], w4 = rp[4];
			a0 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w1 << 32) | w0, b0, a0);
			a1 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w2 << 32) | w1, b1, a1);
			a2 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w3 << 32) | w2, b2, a2);
			a3 = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w4 << 32) | w3, b3, a3);
		} else if (KIND == 7) { // v_mqsad_u32_u8: 4 x u32 accumulators per instruction
			typedef uint32_t u4 __attribute__((ext_vector_type(4)));
			u4 acc = { b0, b1, b2, b3 };
			acc = __builtin_amdgcn_mqsad_u32_u8(((uint64_t)c1 << 32) | c0, c2 | 0x01010101u, acc);
			acc = __builtin_amdgcn_mqsad_u32_u8(((uint64_t)c2 << 32) | c1, c3 | 0x01010101u, acc);
			acc = __builtin_amdgcn_mqsad_u32_u8(((uint64_t)c3 << 32) | c2, c0 | 0x01010101u, acc);
			acc = __builtin_amdgcn_mqsad_u32_u8(((uint64_t)c0 << 32) | c3, c1 | 0x01010101u, acc);
			b0 = acc.x; b1 = acc.y; b2 = acc.z; b3 = acc.w;
		} else if (KIND == 8) { // v_perm_b32 x4
			b0 = __builtin_amdgcn_perm(c0, b0, 0x06050403u);
			b1 = __builtin_amdgcn_perm(c1, b1, 0x05040302u);
			b2 = __builtin_amdgcn_perm(c2, b2, 0x04030201u);
			b3 = __builtin_amdgcn_perm(c3, b3, 0x06050403u);
		} else if (KIND == 9) { // 2 x ds_read_b64 (8-byte aligned)
			const uint8_t *q = lds + (lane & 7) * 8 + (lane >> 3) * 48 + (it & 15) * 48;
			const uint2 v = *(const uint2 *)q, v2 = *(const uint2 *)(q + 8);
			b0 += v.x; b1 += v.y; b2 += v2.x; b3 += v2.y;
		}
	}
	const uint64_t t1 = __builtin_amdgcn_s_memtime();
	out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3) + b0 + b1 + b2 + b3;
	if (lane == 0)
		cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND>
static void run(const char *name, int ops_per_iter)
{
	uint32_t *d_out;
	uint64_t *d_cyc;
	hipMalloc(&d_out, 256 * 1024 * 4);
	hipMalloc(&d_cyc, 256 * 16 * 8);
	printf("%-34s", name);
	for (int wps = 1; wps <= 4; wps *= 2) {
		const int threads = 256 * wps;
		for (int rep = 0; rep < 2; rep++)
			hipLaunchKernelGGL(k_bench<KIND>, dim3(256), dim3(threads), 0, 0, d_out, d_cyc, 12345u + rep);
		hipDeviceSynchronize();
		std::vector<uint64_t> h(256 * 4 * wps);
		hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost);
		double s = 0;
		for (uint64_t v : h)
			s += (double)v;
		s /= h.size();
		printf("  %d w/SIMD: %6.2f cyc/op", wps, s / (double)(N_ITER * ops_per_iter) / wps);
	}
	printf("\n");
	hipFree(d_out);
	hipFree(d_cyc);
}

int main()
{
	printf("SIMD cycles per wave-instruction (lower = faster); ops = the named instruction only\n");
	run<3>("v_add_u32", 4);
	run<1>("v_sad_u8", 4);
	run<0>("v_qsad_pk_u16_u8", 4);
	run<7>("v_mqsad_u32_u8", 4);
	run<2>("v_alignbyte_b32", 4);
	run<8>("v_perm_b32", 4);
	run<4>("5 x ds_read_b32 (per read)", 5);
	run<5>("ds_read_b128 + b32 (per pair)", 1);
	run<9>("2 x ds_read_b64 (per pair)", 1);
	run<6>("k_inter row: 5 rd + 4 qsad (per row)", 1);
	return 0;
}
