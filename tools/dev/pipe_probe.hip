// Does a long kernel (with more work queued behind it) on stream j slow down a chain of short
// kernels on stream 0? (development tool: hardware queues that share a pipe)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_long(unsigned *stop, int us)
{
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < (long long)us * 100) // 100 MHz
		if (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
			break;
}
__global__ void k_short(unsigned *w) { if (threadIdx.x == 0) atomicAdd(w, 1u); }
int main(int argc, char **argv)
{
	const int n = argc > 1 ? atoi(argv[1]) : 8;
	const int behind = argc > 2 ? atoi(argv[2]) : 1;
	const int prio = argc > 3 ? atoi(argv[3]) : 0; // bit 0: the chain's stream at the highest priority; bit 1: the others at the lowest
	int lo = 0, hi = 0;
	hipDeviceGetStreamPriorityRange(&lo, &hi); // (lo = least, hi = greatest: numerically lower is higher)
	printf("stream priorities: least %d, greatest %d; chain stream %s, the others %s\n", lo, hi, prio & 1 ? "greatest" : "default", prio & 2 ? "least" : "default");
	std::vector<hipStream_t> st(n);
	for (int i = 0; i < n; i++) {
		if (i == 0 && (prio & 1))
			hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, hi);
		else if (i > 0 && (prio & 2))
			hipStreamCreateWithPriority(&st[i], hipStreamNonBlocking, lo);
		else
			hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
	}
	unsigned *d;
	hipMalloc(&d, 4096);
	hipMemset(d, 0, 4096);
	auto chain = [&]() {
		hipStreamSynchronize(st[0]);
		auto t0 = std::chrono::steady_clock::now();
		for (int i = 0; i < 100; i++)
			hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, st[0], d + 8);
		hipStreamSynchronize(st[0]);
		return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 100;
	};
	for (int i = 0; i < 3; i++)
		chain();
	printf("baseline: %.2f us per short kernel\n", chain());
	for (int j = 1; j < n; j++) {
		hipMemset(d, 0, 64);
		hipLaunchKernelGGL(k_long, dim3(1), dim3(64), 0, st[j], d, 20000);
		for (int b = 0; b < behind; b++)
			hipLaunchKernelGGL(k_short, dim3(1), dim3(64), 0, st[j], d + 9);
		const double t = chain();
		unsigned one = 1;
		hipMemcpy(d, &one, 4, hipMemcpyHostToDevice);
		hipDeviceSynchronize();
		printf("long kernel on stream %d (+%d queued behind it): %.2f us per short kernel on stream 0\n", j, behind, t);
	}
	return 0;
}
