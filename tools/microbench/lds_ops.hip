// LDS instruction costs on gfx950, as the kernels of this library use them: clocks per wave instruction with 1, 4 and 8 waves of one
// work-group issuing the same pattern (the LDS pipe is shared by the CU's four SIMDs).  Measurement aid:
//   hipcc --offload-arch=gfx950 -O3 -o lds_ops tools/microbench/lds_ops.hip && ./lds_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REPS 256
enum { RD64_UNIT, RD64_BCAST, RD64_ODDROW, RD128_BCAST, RD2x64_BCAST, RD128_UNIT, WR64_UNIT, ADD64_UNIT, ADD64_SAME, NPAT };
static const char* kNames[NPAT] = { "ds_read_b64, lane i -> double i", "ds_read_b64, all lanes one address (broadcast)", "ds_read_b64, lane i -> row i of stride 97 doubles",
	"ds_read_b128, all lanes one address", "ds_read2_b64, all lanes one address", "ds_read_b128, lane i -> 16 bytes i", "ds_write_b64, lane i -> double i",
	"ds_add_f64 (no return), lane i -> double i", "ds_add_f64 (no return), 8 lanes per address" };

template <int PAT>
__global__ void k_lds(unsigned long long* out, double* sink)
{
	extern __shared__ double lds[];
	const int tid = threadIdx.x, lane = tid & 63;
	for (int i = tid; i < 97 * 64 + 64; i += blockDim.x) lds[i] = i * 0.5;
	__syncthreads();
	double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
	const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 8
	for (int r = 0; r < REPS; r++)
	{
		const int o = (r & 7) * 8; // (addresses vary with r so that nothing is hoisted)
		if (PAT == RD64_UNIT) acc0 += lds[lane + o];
		else if (PAT == RD64_BCAST) acc0 += lds[o + 3];
		else if (PAT == RD64_ODDROW) acc0 += lds[lane * 97 + (o >> 3)];
		else if (PAT == RD128_BCAST) { const double2 v = *reinterpret_cast<const double2*>(&lds[o + 2]); acc0 += v.x; acc1 += v.y; }
		else if (PAT == RD2x64_BCAST) { acc0 += lds[o + 1]; acc1 += lds[o + 40]; }
		else if (PAT == RD128_UNIT) { const double2 v = *reinterpret_cast<const double2*>(&lds[2 * lane + o]); acc0 += v.x; acc1 += v.y; }
		else if (PAT == WR64_UNIT) lds[lane + o] = acc0 + r;
		else if (PAT == ADD64_UNIT) __hip_atomic_fetch_add(&lds[lane + o], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
		else if (PAT == ADD64_SAME) __hip_atomic_fetch_add(&lds[(lane >> 3) + o], 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
	}
	__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
	const unsigned long long t1 = __builtin_readcyclecounter();
	if (lane == 0) out[blockIdx.x * 16 + (tid >> 6)] = t1 - t0;
	sink[blockIdx.x * blockDim.x + tid] = acc0 + acc1 + acc2 + acc3 + lds[lane];
}

template <int PAT> static void run(int waves, unsigned long long* d_out, double* d_sink)
{
	hipLaunchKernelGGL(k_lds<PAT>, dim3(1), dim3(64 * waves), (97 * 64 + 64) * sizeof(double), 0, d_out, d_sink);
	hipLaunchKernelGGL(k_lds<PAT>, dim3(1), dim3(64 * waves), (97 * 64 + 64) * sizeof(double), 0, d_out, d_sink);
	(void)hipDeviceSynchronize();
	std::vector<unsigned long long> h(16);
	(void)hipMemcpy(h.data(), d_out, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
	unsigned long long mx = 0;
	for (int w = 0; w < waves; w++) mx = h[w] > mx ? h[w] : mx;
	printf("  %d wave(s): %6.1f clocks per wave instruction (%.1f per instruction of the CU)", waves, (double)mx / REPS, (double)mx / REPS / waves);
}
template <int PAT> static void pattern(unsigned long long* d_out, double* d_sink)
{
	printf("%-52s", kNames[PAT]);
	run<PAT>(1, d_out, d_sink); run<PAT>(4, d_out, d_sink); run<PAT>(8, d_out, d_sink);
	printf("\n");
}
int main()
{
	unsigned long long* d_out; double* d_sink;
	(void)hipMalloc(&d_out, 4096); (void)hipMalloc(&d_sink, 1 << 20);
	pattern<RD64_UNIT>(d_out, d_sink); pattern<RD64_BCAST>(d_out, d_sink); pattern<RD64_ODDROW>(d_out, d_sink); pattern<RD128_BCAST>(d_out, d_sink);
	pattern<RD2x64_BCAST>(d_out, d_sink); pattern<RD128_UNIT>(d_out, d_sink); pattern<WR64_UNIT>(d_out, d_sink); pattern<ADD64_UNIT>(d_out, d_sink);
	pattern<ADD64_SAME>(d_out, d_sink);
	return 0;
}
