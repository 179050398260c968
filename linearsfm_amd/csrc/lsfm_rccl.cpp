// RCCL behind lsfm_allreduce_fn (include/lsfm.h): the all-reduce a C / C++ host hands to lsfm_tree_set_comm for the
// feature-sharded top levels of a tree over several GPUs.  A library of its own (liblsfm_rccl.so,
// linked against librccl) so that liblsfm_hip.so itself links no collective library: a host that brings another transport
// (torch.distributed as linearsfm_amd/distributed.py does, MPI) passes its own function instead.
//
// One communicator per process / GPU.  The reduced arrays live in a device buffer this object owns; every sum is ONE in-place
// ncclAllReduce enqueued on the library's own stream -- ordered on the device, the host does not wait.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstring>
#include <new>

#include "../../include/lsfm.h"
#include "../../include/lsfm_rccl.h"

struct lsfm_rccl {
	ncclComm_t comm = nullptr;
	int rank = 0, world = 1, device = 0;
	char* buf = nullptr;
	size_t bytes = 0;
	long calls = 0;
	double elements = 0;
};

namespace {
int allreduce(void* user, size_t offset_bytes, size_t count, int dtype, void* hip_stream)
{
	lsfm_rccl* c = static_cast<lsfm_rccl*>(user);
	if (!c || !c->comm || offset_bytes + count * 8 > c->bytes) return 1;
	void* p = c->buf + offset_bytes;
	c->calls++; c->elements += (double)count;
	const ncclResult_t r = ncclAllReduce(p, p, count, dtype == LSFM_DTYPE_I64 ? ncclInt64 : ncclDouble, ncclSum, c->comm, static_cast<hipStream_t>(hip_stream));
	if (r != ncclSuccess) { fprintf(stderr, "liblsfm_rccl: ncclAllReduce: %s\n", ncclGetErrorString(r)); return 2; }
	return 0;
}
} // namespace

extern "C" {

int lsfm_rccl_unique_id(void* out, size_t cap)
{
	if (!out || cap < sizeof(ncclUniqueId)) return LSFM_ERR_ARG;
	ncclUniqueId id;
	if (ncclGetUniqueId(&id) != ncclSuccess) return LSFM_ERR_INTERNAL;
	memcpy(out, &id, sizeof id);
	return LSFM_OK;
}
size_t lsfm_rccl_unique_id_bytes(void) { return sizeof(ncclUniqueId); }

int lsfm_rccl_create(const void* unique_id, int rank, int world, int device, size_t buffer_bytes, lsfm_rccl** out)
{
	if (!out || !unique_id || world < 1 || rank < 0 || rank >= world || buffer_bytes < 4096) return LSFM_ERR_ARG;
	*out = nullptr;
	if (hipSetDevice(device) != hipSuccess) return LSFM_ERR_NO_DEVICE;
	lsfm_rccl* c = new (std::nothrow) lsfm_rccl();
	if (!c) return LSFM_ERR_INTERNAL;
	c->rank = rank; c->world = world; c->device = device;
	ncclUniqueId id;
	memcpy(&id, unique_id, sizeof id);
	const ncclResult_t r = ncclCommInitRank(&c->comm, world, id, rank);
	if (r != ncclSuccess)
	{
		fprintf(stderr, "liblsfm_rccl: ncclCommInitRank: %s\n", ncclGetErrorString(r));
		delete c;
		return LSFM_ERR_INTERNAL;
	}
	if (hipMalloc(reinterpret_cast<void**>(&c->buf), buffer_bytes) != hipSuccess)
	{
		ncclCommDestroy(c->comm);
		delete c;
		return LSFM_ERR_HIP;
	}
	(void)hipMemset(c->buf, 0, buffer_bytes);
	c->bytes = buffer_bytes;
	*out = c;
	return LSFM_OK;
}

int lsfm_rccl_attach(lsfm_rccl* c, lsfm_tree* tree)
{
	if (!c || !tree) return LSFM_ERR_ARG;
	return lsfm_tree_set_comm(tree, c->rank, c->world, allreduce, c, c->buf, c->bytes);
}

void lsfm_rccl_counters(const lsfm_rccl* c, long* calls, double* elements)
{
	if (calls) *calls = c ? c->calls : 0;
	if (elements) *elements = c ? c->elements : 0.0;
}

void lsfm_rccl_destroy(lsfm_rccl* c)
{
	if (!c) return;
	(void)hipSetDevice(c->device);
	(void)hipDeviceSynchronize();
	if (c->comm) ncclCommDestroy(c->comm);
	if (c->buf) (void)hipFree(c->buf);
	delete c;
}

} // extern "C"
