/* liblsfm_rccl.so: RCCL as the all-reduce of the feature-sharded top levels (include/lsfm.h, lsfm_tree_set_comm) for a C / C++
 * host.  No counterpart in the reference (one process, one thread: LinearSFMImp.cpp:1938-2033 joins the pairs of a level one
 * after the other).  Optional: liblsfm_hip.so links no collective library; linearsfm_amd/distributed.py passes torch.distributed's
 * all_reduce through the same hook instead.
 *
 *   rank 0:   lsfm_rccl_unique_id(id, sizeof id)            -> hand `id` to every rank by any means (file, socket, MPI_Bcast)
 *   all:      lsfm_rccl_create(id, rank, world, device, bytes, &c)   one communicator + one device buffer per process / GPU
 *             lsfm_rccl_attach(c, top_tree)                  = lsfm_tree_set_comm(tree, rank, world, <ncclAllReduce>, c, buffer, bytes)
 *             lsfm_tree_run(ctx, top_tree, &stats)           every rank, at the same time
 *             lsfm_rccl_destroy(c)
 * bytes: room for the largest camera system of the top levels (288 bytes per 6x6 block of S) + 48 bytes per pose. */
#ifndef LSFM_RCCL_H
#define LSFM_RCCL_H
#include <stddef.h>

#include "lsfm.h"
#ifdef __cplusplus
extern "C" {
#endif

typedef struct lsfm_rccl lsfm_rccl;
size_t lsfm_rccl_unique_id_bytes(void);                 /* sizeof(ncclUniqueId): 128 */
int lsfm_rccl_unique_id(void* out, size_t cap);         /* ncclGetUniqueId */
int lsfm_rccl_create(const void* unique_id, int rank, int world, int device, size_t buffer_bytes, lsfm_rccl** out);
int lsfm_rccl_attach(lsfm_rccl* comm, lsfm_tree* tree);
/* how many sums crossed the GPUs through this communicator so far, and how many 8-byte elements */
void lsfm_rccl_counters(const lsfm_rccl* comm, long* calls, double* elements);
void lsfm_rccl_destroy(lsfm_rccl* comm);

#ifdef __cplusplus
}
#endif
#endif
