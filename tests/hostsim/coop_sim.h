// TEST-ONLY host emulation of the 16-lane row operations of mink_amd/csrc/mk_coop.hip.
//
// The cooperative kernels are written as one program per lane whose lanes meet at row
// operations (broadcast, shift, ballot).  Here the 16 lanes of a row are 16 fibers
// (ucontext) that run the same device function on the CPU; a row operation is an exchange
// point: every fiber deposits its value and yields until all 16 have arrived, then reads the
// values of the others.  Scheduling is round robin and deterministic.
#pragma once
#include <ucontext.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

namespace cosim {
constexpr int N = 16;
struct Sim {
  ucontext_t main_ctx, ctx[N];
  std::vector<char> stack[N];
  bool finished[N];
  int cur;
  double buf[2][N];
  double out[N][N];
  int arrived[2], left[2];
  long epoch[N];
  std::function<void()> body;
};
inline Sim*& sim() {
  static thread_local Sim* s = nullptr;
  return s;
}
inline void yield_next() {
  Sim* S = sim();
  const int from = S->cur;
  for (int k = 1; k < N; ++k) {
    const int n = (from + k) % N;
    if (!S->finished[n]) {
      S->cur = n;
      swapcontext(&S->ctx[from], &S->ctx[n]);
      return;
    }
  }
  fprintf(stderr, "coop_sim: lane %d waits at a row operation that no other lane will reach\n", from);
  abort();
}
inline void trampoline() {
  Sim* S = sim();
  S->body();
  S->finished[S->cur] = true;
  for (int k = 1; k < N; ++k) {
    const int n = (S->cur + k) % N;
    if (!S->finished[n]) {
      S->cur = n;
      setcontext(&S->ctx[n]);
    }
  }
  setcontext(&S->main_ctx);
}
// Runs `body` once per lane of a row, as 16 cooperating fibers.
inline void run_row(const std::function<void()>& body) {
  Sim S;
  S.body = body;
  S.arrived[0] = S.arrived[1] = S.left[0] = S.left[1] = 0;
  for (int i = 0; i < N; ++i) {
    S.finished[i] = false;
    S.epoch[i] = 0;
    S.stack[i].resize(1 << 20);
    getcontext(&S.ctx[i]);
    S.ctx[i].uc_stack.ss_sp = S.stack[i].data();
    S.ctx[i].uc_stack.ss_size = S.stack[i].size();
    S.ctx[i].uc_link = nullptr;
    makecontext(&S.ctx[i], (void (*)())trampoline, 0);
  }
  S.cur = 0;
  Sim* prev = sim();
  sim() = &S;
  swapcontext(&S.main_ctx, &S.ctx[0]);
  sim() = prev;
}
}  // namespace cosim

inline int co_sim_lane() { return cosim::sim()->cur; }
inline const double* co_sim_exchange(double v) {
  cosim::Sim* S = cosim::sim();
  const int f = S->cur;
  const int p = (int)(S->epoch[f]++ & 1);
  S->buf[p][f] = v;
  S->arrived[p]++;
  while (S->arrived[p] < cosim::N) cosim::yield_next();
  memcpy(S->out[f], S->buf[p], sizeof(double) * cosim::N);
  if (++S->left[p] == cosim::N) S->arrived[p] = S->left[p] = 0;
  return S->out[f];
}
