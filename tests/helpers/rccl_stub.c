/* Test double for librccl on a one-GPU box (tests/test_gpu_parity.py::test_bcast_weights_non_root_branch_with_a_stub).
 *
 * nbc_bcast_weights (include/nbc.h) looks ncclBroadcast / ncclCommUserRank up in the host process at call time; a
 * test process that loads THIS library RTLD_GLOBAL before any real RCCL is visible makes the context under test
 * "rank 1 of 2": the broadcast then copies, device to device on the caller's stream, the blob the test planted as
 * rank 0's.  It exists to drive the receiving branch (allocate, receive, attach, free the old blob), which no
 * one-GPU box can reach with a real communicator.  Nothing under neuralbarkcalculator_amd/ links or loads it. */
#include <hip/hip_runtime_api.h>
#include <stddef.h>

static const void* g_root_blob;
static size_t g_root_bytes;
static int g_rank = 1, g_fail, g_calls;
static size_t g_last_count;

void stub_plant_root_blob(const void* dev_blob, size_t bytes) { g_root_blob = dev_blob; g_root_bytes = bytes; }
void stub_set_rank(int rank) { g_rank = rank; }
void stub_fail_next(int code) { g_fail = code; }
int stub_calls(void) { return g_calls; }
size_t stub_last_count(void) { return g_last_count; }

int ncclCommUserRank(void* comm, int* rank) {
  (void)comm;
  *rank = g_rank;
  return 0;
}

int ncclBroadcast(const void* sendbuff, void* recvbuff, size_t count, int datatype, int root, void* comm, hipStream_t stream) {
  (void)sendbuff; (void)comm;
  ++g_calls;
  g_last_count = count;
  if (g_fail) { int f = g_fail; g_fail = 0; return f; }
  if (datatype != 1 /* ncclUint8 */) return 4;
  if (g_rank == root) return 0;                          /* the root's buffer already holds the blob */
  if (!g_root_blob || count > g_root_bytes) return 4;    /* ncclInvalidArgument */
  return hipMemcpyAsync(recvbuff, g_root_blob, count, hipMemcpyDeviceToDevice, stream) == hipSuccess ? 0 : 1;
}
