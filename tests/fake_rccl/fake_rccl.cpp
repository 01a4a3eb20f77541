// TEST INFRASTRUCTURE -- a stand-in for librccl.so that lets SEVERAL rank processes share ONE GPU.
//
// RCCL refuses two ranks on one device, so on the one-GPU boxes the tests have, the library's native exchange
// (dynamite_amd/csrc/comm.cpp) could only ever talk to itself (dnm_comm_loopback).  This file implements the ten RCCL
// entry points comm.cpp binds (rccl_load: DNM_RCCL_LIB names this file) over mailboxes in /dev/shm: every rank is a
// real process with its own handle, its own block of the vector and its own message lists, and what is exercised is
// exactly what a loop-back cannot show -- that the sends one rank posts are the receives its peers post, in the same
// order and of the same sizes (ncclSend / ncclRecv match per ordered pair of ranks, first in first out), that the
// all-gather of windows and the all-reduces of the solver hooks line up, and that nothing waits for a message nobody
// sends.  Sizes are CHECKED here (real RCCL would corrupt or hang): a receive that meets a message of another length
// fails with ncclInvalidUsage, a message that does not arrive within DNM_FAKE_RCCL_TIMEOUT_S (default 120) fails
// with ncclSystemError.  DNM_FAKE_RCCL_FAIL / DNM_FAKE_RCCL_HANG make the communicator fail to come up / the first
// exchange never return (bench.py's first-contact probe is tested against both).
//
// Semantics kept: operations are stream-ordered (everything queued on the stream before the call has completed before
// a message is read from device memory; the call returns with the received data in place, which is stricter than
// RCCL's asynchronous kernels -- the asynchronous hand-overs between the exchange stream and the compute stream are
// what the loop-back tests with the real RCCL cover); a group posts its sends before it waits for any receive, so a
// pair of ranks that send to each other inside one group cannot deadlock, exactly as with ncclGroupStart/End.
//
// Not a product transport: host-staged copies, one file per message.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dirent.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include <cerrno>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

struct Op {
  bool send;
  const void *src;
  void *dst;
  size_t bytes;
  int peer;
  hipStream_t stream;
};

struct Comm {
  std::string dir;
  int rank = 0, nranks = 1;
  std::vector<uint64_t> sent, rcvd;        // messages so far per peer
  std::vector<char> host;
};

// DNM_FAKE_RCCL_STATS=1: where the time of this transport went, printed when the communicator is destroyed
struct Stats { double sync = 0, copy = 0, put = 0, wait = 0; long groups = 0, msgs = 0, coll = 0; } g_stats;

thread_local int g_depth = 0;
thread_local std::vector<std::pair<Comm *, Op>> g_queue;
thread_local char g_err[256] = "no error";

// DNM_FAKE_RCCL_HOST=1: every buffer is host memory and there is no device (the CPU tests of this file itself)
bool host_only() {
  static const bool v = getenv("DNM_FAKE_RCCL_HOST") != nullptr;
  return v;
}
hipError_t copy(void *dst, const void *src, size_t bytes) {
  if (host_only()) {
    memcpy(dst, src, bytes);
    return hipSuccess;
  }
  return hipMemcpy(dst, src, bytes, hipMemcpyDefault);
}
hipError_t sync(hipStream_t s) { return host_only() ? hipSuccess : hipStreamSynchronize(s); }

double now_s() {
  timeval tv;
  gettimeofday(&tv, nullptr);
  return tv.tv_sec + 1e-6 * tv.tv_usec;
}

double timeout_s() {
  const char *e = getenv("DNM_FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 120.0;
}

ncclResult_t fail(ncclResult_t r, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
ncclResult_t fail(ncclResult_t r, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  fprintf(stderr, "[fake_rccl] %s\n", g_err);
  return r;
}

std::string msg_path(const Comm *c, int src, int dst, uint64_t seq) {
  char buf[96];
  snprintf(buf, sizeof buf, "/m_%d_%d_%llu", src, dst, (unsigned long long)seq);
  return c->dir + buf;
}

ncclResult_t put(Comm *c, int peer, const void *host, size_t bytes) {
  const std::string final_path = msg_path(c, c->rank, peer, c->sent[(size_t)peer]++);
  const std::string tmp = final_path + ".tmp";
  int fd = open(tmp.c_str(), O_CREAT | O_WRONLY | O_TRUNC, 0600);
  if (fd < 0) return fail(ncclSystemError, "cannot create %s", tmp.c_str());
  size_t done = 0;
  while (done < bytes) {
    ssize_t w = write(fd, (const char *)host + done, bytes - done);
    if (w < 0) {
      if (errno == EINTR) continue;
      close(fd);
      return fail(ncclSystemError, "write to %s failed", tmp.c_str());
    }
    done += (size_t)w;
  }
  close(fd);
  if (rename(tmp.c_str(), final_path.c_str()) != 0) return fail(ncclSystemError, "rename of %s failed", tmp.c_str());
  return ncclSuccess;
}

ncclResult_t get(Comm *c, int peer, void *host, size_t bytes) {
  const std::string path = msg_path(c, peer, c->rank, c->rcvd[(size_t)peer]++);
  const double t0 = now_s(), limit = timeout_s();
  int fd = -1;
  while ((fd = open(path.c_str(), O_RDONLY)) < 0) {
    const double waited = now_s() - t0;
    if (waited > limit)
      return fail(ncclSystemError, "rank %d: message %llu from rank %d never arrived (a receive nobody sends to)", c->rank,
                  (unsigned long long)(c->rcvd[(size_t)peer] - 1), peer);
    if (waited < 2e-3) sched_yield();            // the peer is usually a few microseconds behind
    else usleep(waited < 0.1 ? 100 : 1000);
  }
  struct stat st;
  fstat(fd, &st);
  if ((size_t)st.st_size != bytes) {
    close(fd);
    return fail(ncclInvalidUsage, "rank %d, message %llu from rank %d: the sender posted %ld bytes, the receiver %ld", c->rank,
                (unsigned long long)(c->rcvd[(size_t)peer] - 1), peer, (long)st.st_size, (long)bytes);
  }
  size_t done = 0;
  while (done < bytes) {
    ssize_t r = read(fd, (char *)host + done, bytes - done);
    if (r < 0 && errno == EINTR) continue;
    if (r <= 0) {
      close(fd);
      return fail(ncclSystemError, "short read of %s", path.c_str());
    }
    done += (size_t)r;
  }
  close(fd);
  unlink(path.c_str());
  return ncclSuccess;
}

size_t type_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    default: return 8;
  }
}

// everything queued so far: sends first (they never block), then the receives in posting order
ncclResult_t run(std::vector<std::pair<Comm *, Op>> &queue) {
  std::vector<std::pair<Comm *, Op>> q;
  q.swap(queue);                                 // (a failure leaves nothing behind for the next group)
  if (getenv("DNM_FAKE_RCCL_HANG")) {            // tests: an exchange that never completes
    for (;;) sleep(1000);
  }
  std::vector<hipStream_t> synced;
  ++g_stats.groups;
  double t0 = now_s();
  for (auto &e : q) {
    bool seen = false;
    for (hipStream_t s : synced) seen = seen || s == e.second.stream;
    if (!seen) {
      if (sync(e.second.stream) != hipSuccess) return fail(ncclUnhandledCudaError, "hipStreamSynchronize failed");
      synced.push_back(e.second.stream);
    }
  }
  g_stats.sync += now_s() - t0;
  for (auto &e : q) {
    Comm *c = e.first;
    const Op &o = e.second;
    if (!o.send) continue;
    ++g_stats.msgs;
    if (c->host.size() < o.bytes) c->host.resize(o.bytes);
    t0 = now_s();
    if (o.bytes && copy(c->host.data(), o.src, o.bytes) != hipSuccess)
      return fail(ncclUnhandledCudaError, "copy of a send buffer to the host failed");
    g_stats.copy += now_s() - t0;
    t0 = now_s();
    ncclResult_t r = put(c, o.peer, c->host.data(), o.bytes);
    g_stats.put += now_s() - t0;
    if (r != ncclSuccess) return r;
  }
  for (auto &e : q) {
    Comm *c = e.first;
    const Op &o = e.second;
    if (o.send) continue;
    if (c->host.size() < o.bytes) c->host.resize(o.bytes);
    t0 = now_s();
    ncclResult_t r = get(c, o.peer, c->host.data(), o.bytes);
    g_stats.wait += now_s() - t0;
    if (r != ncclSuccess) return r;
    t0 = now_s();
    if (o.bytes && copy(o.dst, c->host.data(), o.bytes) != hipSuccess)
      return fail(ncclUnhandledCudaError, "copy of a received message to its buffer failed");
    g_stats.copy += now_s() - t0;
  }
  return ncclSuccess;
}

ncclResult_t post(Comm *c, const Op &o) {
  if (o.peer < 0 || o.peer >= c->nranks) return fail(ncclInvalidArgument, "peer out of range");
  g_queue.emplace_back(c, o);
  return g_depth > 0 ? ncclSuccess : run(g_queue);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof *id);
  unsigned long long r = 0;
  FILE *f = fopen("/dev/urandom", "rb");
  if (f) {
    if (fread(&r, sizeof r, 1, f) != 1) r = (unsigned long long)getpid() * 2654435761ull;
    fclose(f);
  }
  snprintf(id->internal, sizeof id->internal, "dnmfake_%d_%016llx", (int)getpid(), r);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank) {
  if (getenv("DNM_FAKE_RCCL_FAIL")) return fail(ncclSystemError, "DNM_FAKE_RCCL_FAIL is set (tests: a transport that does not come up)");
  Comm *c = new Comm();
  id.internal[sizeof id.internal - 1] = 0;
  c->dir = std::string("/dev/shm/") + id.internal;
  c->rank = rank;
  c->nranks = nranks;
  c->sent.assign((size_t)nranks, 0);
  c->rcvd.assign((size_t)nranks, 0);
  if (mkdir(c->dir.c_str(), 0700) != 0 && errno != EEXIST) {
    delete c;
    return fail(ncclSystemError, "cannot create %s", c->dir.c_str());
  }
  // every rank greets every other one: the communicator exists once all have arrived
  char hello = 1;
  for (int q = 0; q < nranks; ++q)
    if (q != rank && put(c, q, &hello, 1) != ncclSuccess) return ncclSystemError;
  for (int q = 0; q < nranks; ++q)
    if (q != rank && get(c, q, &hello, 1) != ncclSuccess) return ncclSystemError;
  *out = (ncclComm_t)c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm *c = (Comm *)comm;
  if (!c) return ncclSuccess;
  if (getenv("DNM_FAKE_RCCL_STATS"))
    fprintf(stderr, "[fake_rccl] rank %d: %ld groups, %ld messages sent, %ld all-reduces; stream syncs %.2f s, copies %.2f s, "
            "writing %.2f s, waiting for messages %.2f s\n", c->rank, g_stats.groups, g_stats.msgs, g_stats.coll, g_stats.sync,
            g_stats.copy, g_stats.put, g_stats.wait);
  // what this rank never read (a failed test) goes with it; the directory with the last rank to leave
  if (DIR *d = opendir(c->dir.c_str())) {
    while (dirent *e = readdir(d)) {
      int src = -1, dst = -1;
      if (sscanf(e->d_name, "m_%d_%d_", &src, &dst) == 2 && dst == c->rank) unlink((c->dir + "/" + e->d_name).c_str());
    }
    closedir(d);
  }
  rmdir(c->dir.c_str());
  delete c;
  return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : g_err; }

ncclResult_t ncclGroupStart() {
  ++g_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return fail(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
  if (--g_depth > 0) return ncclSuccess;
  return run(g_queue);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  return post((Comm *)comm, Op{true, buf, nullptr, count * type_size(t), peer, s});
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  return post((Comm *)comm, Op{false, nullptr, buf, count * type_size(t), peer, s});
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
  Comm *c = (Comm *)comm;
  const size_t bytes = count * type_size(t);
  if (g_depth > 0) return fail(ncclInvalidUsage, "collectives inside a group are not part of this stand-in");
  std::vector<std::pair<Comm *, Op>> q;
  for (int p = 0; p < c->nranks; ++p)
    if (p != c->rank) q.emplace_back(c, Op{true, send, nullptr, bytes, p, s});
  for (int p = 0; p < c->nranks; ++p)
    if (p != c->rank) q.emplace_back(c, Op{false, nullptr, (char *)recv + (size_t)p * bytes, bytes, p, s});
  ncclResult_t r = run(q);
  if (r != ncclSuccess) return r;
  // (device to device: hipMemcpy would return before the copy has run, and the caller's stream does not wait for the
  // null stream -- on the collective's own stream, then)
  if (bytes && !host_only() &&
      (hipMemcpyAsync((char *)recv + (size_t)c->rank * bytes, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess || sync(s) != hipSuccess))
    return fail(ncclUnhandledCudaError, "all-gather: copy of the rank's own part failed");
  if (bytes && host_only()) memcpy((char *)recv + (size_t)c->rank * bytes, send, bytes);
  return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t s) {
  Comm *c = (Comm *)comm;
  if (t != ncclDouble || (op != ncclSum && op != ncclMax)) return fail(ncclInvalidArgument, "all-reduce: doubles, sum or max");
  if (g_depth > 0) return fail(ncclInvalidUsage, "collectives inside a group are not part of this stand-in");
  const size_t bytes = count * 8;
  ++g_stats.coll;
  if (sync(s) != hipSuccess) return fail(ncclUnhandledCudaError, "hipStreamSynchronize failed");
  std::vector<double> mine(count), other(count);
  if (bytes && copy(mine.data(), send, bytes) != hipSuccess)
    return fail(ncclUnhandledCudaError, "all-reduce: copy to the host failed");
  for (int p = 0; p < c->nranks; ++p)
    if (p != c->rank) {
      ncclResult_t r = put(c, p, mine.data(), bytes);
      if (r != ncclSuccess) return r;
    }
  // summed in rank order on every rank: the same bits everywhere
  std::vector<double> acc(count, 0.0);
  for (int p = 0; p < c->nranks; ++p) {
    const double *v = mine.data();
    if (p != c->rank) {
      ncclResult_t r = get(c, p, other.data(), bytes);
      if (r != ncclSuccess) return r;
      v = other.data();
    }
    for (size_t i = 0; i < count; ++i) acc[i] = (p == 0) ? v[i] : (op == ncclSum ? acc[i] + v[i] : (v[i] > acc[i] ? v[i] : acc[i]));
  }
  if (bytes && copy(recv, acc.data(), bytes) != hipSuccess)
    return fail(ncclUnhandledCudaError, "all-reduce: copy of the result failed");
  return ncclSuccess;
}

}  // extern "C"
