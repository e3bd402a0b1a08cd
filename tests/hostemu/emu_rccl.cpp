// TEST INFRASTRUCTURE -- see shim/rccl/rccl.h
#include <rccl/rccl.h>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>
#include <vector>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>
#include <unistd.h>

// Ranks in DIFFERENT processes (bench.py --gpus N --backend gloo under torch.distributed.run, tests/test_multirank_host.py): with the
// environment variable BLOM_HOSTEMU_RCCL_DIR set, a message is a file <dir>/<key>_<src>_<dst>_<seq> (written under a temporary name,
// then renamed), the receiver polls for the next sequence number of its (src, dst) pair, reads and removes it.
namespace {
const char *proc_dir() { static const char *d = getenv("BLOM_HOSTEMU_RCCL_DIR"); return d && *d ? d : nullptr; }
std::map<std::pair<int, int>, unsigned long long> g_seq_out, g_seq_in;      // per process: messages sent / received per (src, dst)
std::string msg_path(unsigned long long key, int src, int dst, unsigned long long seq) {
  char b[512];
  snprintf(b, sizeof b, "%s/%llu_%d_%d_%llu", proc_dir(), key, src, dst, seq);
  return b;
}
struct World {
  std::mutex mu;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<std::vector<char>>> box;   // (src, dst) -> messages in order
  int refs = 0;
};
std::mutex g_mu;
std::map<unsigned long long, World *> g_worlds;
std::atomic<unsigned long long> g_next{1};
struct Pending { bool send; void *buf; size_t bytes; int peer; };
thread_local std::vector<Pending> t_group;
thread_local int t_depth = 0;
}  // namespace
struct hostemu_comm { World *w; int rank, nranks; unsigned long long key; };

const char *ncclGetErrorString(ncclResult_t) { return "hostemu rccl error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
  memset(id, 0, sizeof(*id));
  const unsigned long long k = proc_dir() ? ((unsigned long long)getpid() << 20) + g_next++ : g_next++;
  memcpy(id->internal, &k, sizeof(k));
  return ncclSuccess;
}
ncclResult_t ncclCommInitRank(ncclComm_t *c, int nranks, ncclUniqueId id, int rank) {
  unsigned long long k;
  memcpy(&k, id.internal, sizeof(k));
  std::lock_guard<std::mutex> l(g_mu);
  World *&w = g_worlds[k];
  if (!w) w = new World();
  w->refs++;
  *c = new hostemu_comm{w, rank, nranks, k};
  return ncclSuccess;
}
ncclResult_t ncclCommDestroy(ncclComm_t c) {
  std::lock_guard<std::mutex> l(g_mu);
  if (--c->w->refs == 0) { g_worlds.erase(c->key); delete c->w; }
  delete c;
  return ncclSuccess;
}
static void do_send(ncclComm_t c, const Pending &p) {
  if (proc_dir()) {
    const unsigned long long seq = g_seq_out[{c->rank, p.peer}]++;
    const std::string f = msg_path(c->key, c->rank, p.peer, seq), tmp = f + ".tmp";
    FILE *fp = fopen(tmp.c_str(), "wb");
    if (fp) { fwrite(p.buf, 1, p.bytes, fp); fclose(fp); rename(tmp.c_str(), f.c_str()); }
    return;
  }
  World &w = *c->w;
  std::vector<char> m((const char *)p.buf, (const char *)p.buf + p.bytes);
  { std::lock_guard<std::mutex> l(w.mu); w.box[{c->rank, p.peer}].push_back(std::move(m)); }
  w.cv.notify_all();
}
static ncclResult_t do_recv(ncclComm_t c, const Pending &p) {
  if (proc_dir()) {
    const unsigned long long seq = g_seq_in[{p.peer, c->rank}]++;
    const std::string f = msg_path(c->key, p.peer, c->rank, seq);
    for (int spin = 0; spin < 60000; spin++) {                      // up to 60 s
      FILE *fp = fopen(f.c_str(), "rb");
      if (fp) {
        const size_t got = fread(p.buf, 1, p.bytes, fp);
        char extra;
        const bool more = fread(&extra, 1, 1, fp) == 1;
        fclose(fp);
        remove(f.c_str());
        if (got != p.bytes || more) {
          fprintf(stderr, "hostemu rccl: rank %d expects %zu bytes from rank %d, the message has %s\n", c->rank, p.bytes, p.peer, more ? "more" : "fewer");
          return ncclInternalError;
        }
        return ncclSuccess;
      }
      std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    fprintf(stderr, "hostemu rccl: rank %d waited 60 s for message %llu of %zu bytes from rank %d\n", c->rank, seq, p.bytes, p.peer);
    return ncclInternalError;
  }
  World &w = *c->w;
  std::unique_lock<std::mutex> l(w.mu);
  auto &q = w.box[{p.peer, c->rank}];
  if (!w.cv.wait_for(l, std::chrono::seconds(20), [&] { return !q.empty(); })) {
    fprintf(stderr, "hostemu rccl: rank %d waited 20 s for a message of %zu bytes from rank %d\n", c->rank, p.bytes, p.peer);
    return ncclInternalError;
  }
  if (q.front().size() != p.bytes) {                              // the two sides disagree on the message size
    fprintf(stderr, "hostemu rccl: rank %d expects %zu bytes from rank %d, the message has %zu\n", c->rank, p.bytes, p.peer, q.front().size());
    return ncclInternalError;
  }
  memcpy(p.buf, q.front().data(), p.bytes);
  q.pop_front();
  return ncclSuccess;
}
thread_local ncclComm_t t_comm = nullptr;
ncclResult_t ncclGroupStart() { t_depth++; return ncclSuccess; }
ncclResult_t ncclGroupEnd() {
  if (--t_depth > 0) return ncclSuccess;
  ncclResult_t rc = ncclSuccess;
  for (auto &p : t_group) if (p.send) do_send(t_comm, p);           // all sends first: a group cannot deadlock
  for (auto &p : t_group) if (!p.send) { ncclResult_t r = do_recv(t_comm, p); if (r) rc = r; }
  t_group.clear();
  return rc;
}
ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t) {
  t_comm = c;
  Pending p{true, (void *)buf, count * 8, peer};
  if (t_depth > 0) { t_group.push_back(p); return ncclSuccess; }
  do_send(c, p);
  return ncclSuccess;
}
ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t, int peer, ncclComm_t c, hipStream_t) {
  t_comm = c;
  Pending p{false, buf, count * 8, peer};
  if (t_depth > 0) { t_group.push_back(p); return ncclSuccess; }
  return do_recv(c, p);
}
