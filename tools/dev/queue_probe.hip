// How many HIP streams of one process really run kernels side by side? (development tool)
// N-1 streams each get a kernel that spins until a flag is set; stream N-1 then sets it.
// If the setter is never reached while the waiters are resident, the waiters time out.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_wait(unsigned *w, unsigned *res, int id)
{
	for (int i = 0; i < (1 << 16); i++) {
		if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
			res[id] = 1;
			return;
		}
		__builtin_amdgcn_s_sleep(64);
	}
	res[id] = 0;
}
__global__ void k_set(unsigned *w) { __hip_atomic_store(w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
int main(int argc, char **argv)
{
	const int maxn = argc > 1 ? atoi(argv[1]) : 10;
	std::vector<hipStream_t> st(maxn);
	for (auto &s : st)
		hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
	unsigned *d;
	hipMalloc(&d, 4096);
	for (int n = 2; n <= maxn; n++) {
		hipMemset(d, 0, 4096);
		for (int i = 0; i < n - 1; i++)
			hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st[i], d, d + 16, i);
		hipLaunchKernelGGL(k_set, dim3(1), dim3(1), 0, st[n - 1], d);
		hipDeviceSynchronize();
		unsigned h[64];
		hipMemcpy(h, d + 16, sizeof(h), hipMemcpyDeviceToHost);
		int ok = 0;
		for (int i = 0; i < n - 1; i++)
			ok += h[i];
		printf("%d streams: %d of %d waiters saw the flag\n", n, ok, n - 1);
	}
	return 0;
}
