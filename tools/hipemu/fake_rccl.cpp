// fake_rccl.cpp — a socket-backed stand-in for the nine RCCL entry points tf_exchange.hip binds, for the lock-step CPU
// emulator only (TEST INFRASTRUCTURE: tools/hipemu/build.py builds it as _build/libfakerccl.so and the world_size-2
// emulator tests point TFGPU_RCCL_LIB at it; the product library never links it).  "Device" memory under the emulator
// is host memory, streams are synchronous, so a grouped send/recv is: one writer thread pushing this rank's sends to
// each peer in call order while the calling thread drains the receives in call order.
//
// Rendezvous: the unique id holds a directory-less path prefix under /tmp; rank r listens on <prefix>.<r>, connects to
// every lower rank and accepts every higher one.
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/un.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Op { bool send; int peer; const char *sp; char *rp; size_t n; };
struct Comm {
  int rank = 0, world = 1, lfd = -1;
  std::string base;
  std::vector<int> fd;  // per peer
};
thread_local int g_depth = 0;
thread_local std::vector<std::pair<Comm *, Op>> g_ops;
const char *g_err = "fake rccl: ok";

bool write_all(int fd, const char *p, size_t n) {
  while (n) { ssize_t k = ::write(fd, p, n); if (k < 0) { if (errno == EINTR) continue; return false; } p += k; n -= (size_t)k; }
  return true;
}
bool read_all(int fd, char *p, size_t n) {
  while (n) { ssize_t k = ::read(fd, p, n); if (k < 0) { if (errno == EINTR) continue; return false; } if (k == 0) return false; p += k; n -= (size_t)k; }
  return true;
}
sockaddr_un addr_of(const std::string &path) {
  sockaddr_un a{};
  a.sun_family = AF_UNIX;
  std::snprintf(a.sun_path, sizeof a.sun_path, "%s", path.c_str());
  return a;
}

int run_ops(std::vector<std::pair<Comm *, Op>> &ops) {
  // self moves first: the k-th send to self pairs with the k-th receive from self
  std::vector<Op *> ss, sr;
  for (auto &e : ops) if (e.second.peer == e.first->rank) (e.second.send ? ss : sr).push_back(&e.second);
  if (ss.size() != sr.size()) { g_err = "fake rccl: unmatched self send/recv"; return 1; }
  for (size_t i = 0; i < ss.size(); i++) {
    if (ss[i]->n != sr[i]->n) { g_err = "fake rccl: self send/recv sizes differ"; return 1; }
    std::memmove(sr[i]->rp, ss[i]->sp, ss[i]->n);
  }
  bool ok_w = true, ok_r = true;
  std::thread writer([&] {
    for (auto &e : ops) if (e.second.send && e.second.peer != e.first->rank) ok_w &= write_all(e.first->fd[(size_t)e.second.peer], e.second.sp, e.second.n);
  });
  for (auto &e : ops) if (!e.second.send && e.second.peer != e.first->rank) ok_r &= read_all(e.first->fd[(size_t)e.second.peer], e.second.rp, e.second.n);
  writer.join();
  if (!ok_w || !ok_r) { g_err = "fake rccl: socket transfer failed (a peer died?)"; return 1; }
  return 0;
}

}  // namespace

extern "C" {

struct ncclUniqueId { char internal[128]; };

int ncclGetUniqueId(ncclUniqueId *id) {
  std::memset(id, 0, sizeof *id);
  unsigned r = 0;
  FILE *f = std::fopen("/dev/urandom", "rb");
  if (f) { if (std::fread(&r, sizeof r, 1, f) != 1) r = 0; std::fclose(f); }
  std::snprintf(id->internal, sizeof id->internal, "/tmp/tfgpu_fakerccl_%d_%08x", (int)getpid(), r);
  return 0;
}

int ncclCommInitRank(void **out, int world, ncclUniqueId id, int rank) {
  auto c = new Comm;
  c->rank = rank; c->world = world; c->base = id.internal;
  c->fd.assign((size_t)world, -1);
  if (world > 1) {
    std::string mine = c->base + "." + std::to_string(rank);
    c->lfd = ::socket(AF_UNIX, SOCK_STREAM, 0);
    sockaddr_un a = addr_of(mine);
    ::unlink(mine.c_str());
    if (c->lfd < 0 || ::bind(c->lfd, (sockaddr *)&a, sizeof a) != 0 || ::listen(c->lfd, world) != 0) { g_err = "fake rccl: bind/listen failed"; return 1; }
    for (int p = 0; p < rank; p++) {  // connect to the lower ranks (they may not be listening yet)
      sockaddr_un pa = addr_of(c->base + "." + std::to_string(p));
      int fd = -1;
      for (int tries = 0; tries < 3000; tries++) {
        fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
        if (::connect(fd, (sockaddr *)&pa, sizeof pa) == 0) break;
        ::close(fd); fd = -1;
        ::usleep(10000);
      }
      if (fd < 0) { g_err = "fake rccl: peer never listened"; return 1; }
      int32_t me = rank;
      if (!write_all(fd, (const char *)&me, 4)) { g_err = "fake rccl: hello failed"; return 1; }
      c->fd[(size_t)p] = fd;
    }
    for (int k = rank + 1; k < world; k++) {  // accept the higher ranks, whoever comes first
      int fd = ::accept(c->lfd, nullptr, nullptr);
      int32_t who = -1;
      if (fd < 0 || !read_all(fd, (char *)&who, 4) || who <= rank || who >= world) { g_err = "fake rccl: accept failed"; return 1; }
      c->fd[(size_t)who] = fd;
    }
    ::close(c->lfd); c->lfd = -1;
    ::unlink(mine.c_str());
  }
  *out = c;
  return 0;
}

int ncclCommDestroy(void *p) {
  auto c = static_cast<Comm *>(p);
  if (!c) return 0;
  for (int fd : c->fd) if (fd >= 0) ::close(fd);
  delete c;
  return 0;
}

int ncclGroupStart() { g_depth++; return 0; }
int ncclGroupEnd() {
  if (--g_depth > 0) return 0;
  auto ops = std::move(g_ops);
  g_ops.clear();
  return run_ops(ops);
}
int ncclSend(const void *buf, size_t count, int dtype, int peer, void *comm, void *) {
  if (dtype != 0 && dtype != 1) { g_err = "fake rccl: only byte types"; return 1; }
  g_ops.push_back({static_cast<Comm *>(comm), Op{true, peer, static_cast<const char *>(buf), nullptr, count}});
  if (g_depth == 0) { auto ops = std::move(g_ops); g_ops.clear(); return run_ops(ops); }
  return 0;
}
int ncclRecv(void *buf, size_t count, int dtype, int peer, void *comm, void *) {
  if (dtype != 0 && dtype != 1) { g_err = "fake rccl: only byte types"; return 1; }
  g_ops.push_back({static_cast<Comm *>(comm), Op{false, peer, nullptr, static_cast<char *>(buf), count}});
  if (g_depth == 0) { auto ops = std::move(g_ops); g_ops.clear(); return run_ops(ops); }
  return 0;
}
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, void *) {
  if (dtype != 0 && dtype != 1) { g_err = "fake rccl: only byte types"; return 1; }
  auto c = static_cast<Comm *>(comm);
  std::vector<std::pair<Comm *, Op>> ops;
  for (int p = 0; p < c->world; p++) {
    ops.push_back({c, Op{true, p, static_cast<const char *>(send), nullptr, count}});
    ops.push_back({c, Op{false, p, nullptr, static_cast<char *>(recv) + (size_t)p * count, count}});
  }
  return run_ops(ops);
}
const char *ncclGetErrorString(int) { return g_err; }

}  // extern "C"
