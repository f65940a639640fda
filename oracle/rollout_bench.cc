// CPU baseline harness (TEST INFRASTRUCTURE, linked against the compiled reference oracle/_ref/liboracle_fast.so;
// never part of the product): the reference engine stepping a batch of rollouts on host threads with the
// SAME initial states and per-step controls the GPU bench uses, i.e. the work of the reference's
// _unsafe_rollout_threaded (python/mujoco/rollout.cc:181-216: a pool of threads, one mjData per thread, each
// rollout = mj_setState + nstep x (set control, mj_step, mj_getState)), restated here as a plain thread team
// over contiguous slices of the batch.
//
//   rollout_bench <model.mjb> <state0.bin> <ctrl.bin> <nroll> <nstep> <nthread> <solver|-1> <integrator|-1> [out_final.bin [warm0.bin]]
//
// state0.bin: [nroll][nstate] doubles (mjSTATE_FULLPHYSICS); ctrl.bin: [nroll][nstep][nu] doubles; warm0.bin (optional):
// [nroll][nv] doubles, the rollouts' initial qacc_warmstart (the rollout API's initial_warmstart).
// $ROLLOUT_BENCH_MIN_S (default 0): after the first pass over the batch, the whole batch is stepped again `repeats` times
// inside ONE thread team so that the timed work lasts at least that many seconds (a 60 ms pass on 256 threads measures
// thread start-up, not mj_step); the rate printed is that of the repeated region, the means are per env-step.
// Prints one line: env_steps_per_s=<v> seconds=<t> repeats=<r> nroll=<n> nstep=<k> nthread=<c> mean_ncon=.. mean_nefc=.. mean_niter=..
#include <mujoco/mujoco.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static std::vector<double> read_doubles(const char* path, size_t n) {
  std::vector<double> v(n);
  FILE* f = fopen(path, "rb");
  if (!f || fread(v.data(), sizeof(double), n, f) != n) { fprintf(stderr, "rollout_bench: cannot read %zu doubles from %s\n", n, path); exit(2); }
  fclose(f);
  return v;
}

int main(int argc, char** argv) {
  if (argc < 9) { fprintf(stderr, "usage: rollout_bench model.mjb state0.bin ctrl.bin nroll nstep nthread solver integrator [out.bin]\n"); return 2; }
  mjModel* m = mj_loadModel(argv[1], nullptr);
  if (!m) { fprintf(stderr, "rollout_bench: cannot load %s\n", argv[1]); return 2; }
  const int nroll = atoi(argv[4]), nstep = atoi(argv[5]);
  int nthread = atoi(argv[6]);
  if (atoi(argv[7]) >= 0) m->opt.solver = atoi(argv[7]);
  if (atoi(argv[8]) >= 0) m->opt.integrator = atoi(argv[8]);
  const int nstate = mj_stateSize(m, mjSTATE_FULLPHYSICS), nu = m->nu;
  const std::vector<double> state0 = read_doubles(argv[2], (size_t)nroll*nstate);
  const std::vector<double> ctrl = read_doubles(argv[3], (size_t)nroll*nstep*nu);
  std::vector<double> warm0;
  if (argc > 10) warm0 = read_doubles(argv[10], (size_t)nroll*m->nv);
  std::vector<double> final_state((size_t)nroll*nstate);
  if (nthread < 1) nthread = 1;
  if (nthread > nroll) nthread = nroll;
  std::vector<mjData*> data(nthread);
  for (auto& d : data) d = mj_makeData(m);
  std::vector<double> acc((size_t)nthread*3, 0.0);

  int repeats = 1;
  auto worker = [&](int t) {
    mjData* d = data[t];
    const int lo = (int)((long long)nroll*t/nthread), hi = (int)((long long)nroll*(t + 1)/nthread);
    for (int rep = 0; rep < repeats; rep++)
    for (int r = lo; r < hi; r++) {
      mj_resetData(m, d);                       // (cold warm start, like a rollout without initial_warmstart)
      mj_setState(m, d, state0.data() + (size_t)r*nstate, mjSTATE_FULLPHYSICS);
      if (!warm0.empty()) mju_copy(d->qacc_warmstart, warm0.data() + (size_t)r*m->nv, m->nv);
      for (int k = 0; k < nstep; k++) {
        mju_copy(d->ctrl, ctrl.data() + ((size_t)r*nstep + k)*nu, nu);
        mj_step(m, d);
        acc[3*t] += d->ncon; acc[3*t + 1] += d->nefc; acc[3*t + 2] += d->solver_niter[0];
      }
      mj_getState(m, d, final_state.data() + (size_t)r*nstate, mjSTATE_FULLPHYSICS);
    }
  };
  auto pass = [&]() {
    for (auto& a : acc) a = 0.0;
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> team;
    for (int t = 0; t < nthread; t++) team.emplace_back(worker, t);
    for (auto& th : team) th.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  double sec = pass();
  const char* min_s = getenv("ROLLOUT_BENCH_MIN_S");
  if (min_s && atof(min_s) > sec) {
    double r = atof(min_s)/(sec > 1e-6 ? sec : 1e-6);
    repeats = r > 100000.0 ? 100000 : (int)r + 1;
    sec = pass();
  }
  double ncon = 0, nefc = 0, niter = 0;
  for (int t = 0; t < nthread; t++) { ncon += acc[3*t]; nefc += acc[3*t + 1]; niter += acc[3*t + 2]; }
  const double total = (double)nroll*nstep*repeats;
  printf("env_steps_per_s=%.1f seconds=%.4f repeats=%d nroll=%d nstep=%d nthread=%d mean_ncon=%.3f mean_nefc=%.3f mean_niter=%.3f\n",
         total/sec, sec, repeats, nroll, nstep, nthread, ncon/total, nefc/total, niter/total);
  if (argc > 9) {
    FILE* f = fopen(argv[9], "wb");
    if (f) { fwrite(final_state.data(), sizeof(double), final_state.size(), f); fclose(f); }
  }
  for (auto& d : data) mj_deleteData(d);
  mj_deleteModel(m);
  return 0;
}
