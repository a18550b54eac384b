// multi_gpu -- one batch of acrobot problems over several devices of a node from ONE C++ process:
//   ./multi_gpu x0.bin B T n_shards iters out.bin [device,device,...]
// x0.bin: B x 4 doubles.  Shard i runs on the i-th listed device (default: device i % #devices-visible... all on device 0 when
// only one is visible: N logical shards on one GPU give the bits N GPUs give).  out.bin: B costs (gathered: RCCL all-gather
// between distinct devices, copies otherwise), then B x T controls.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "ilqr_amd.hpp"

using namespace ilqr_amd;

int main(int argc, char* argv[]) {
  if (argc < 7) {
    std::cerr << "usage: multi_gpu x0.bin B T n_shards iters out.bin [devices]" << std::endl;
    return 1;
  }
  const int B = std::atoi(argv[2]), T = std::atoi(argv[3]), ns = std::atoi(argv[4]), iters = std::atoi(argv[5]);
  std::vector<int> devices(ns, 0);
  if (argc > 7) {
    std::string s(argv[7]);
    size_t pos = 0;
    for (int i = 0; i < ns && pos != std::string::npos; i++) {
      devices[i] = std::atoi(s.c_str() + pos);
      pos = s.find(',', pos);
      if (pos != std::string::npos) pos++;
    }
  }
  try {
    std::vector<double> x0((size_t)B * 4), u0((size_t)B * T, 0.0);
    FILE* f = std::fopen(argv[1], "rb");
    if (!f || std::fread(x0.data(), sizeof(double), x0.size(), f) != x0.size()) throw std::runtime_error("cannot read x0");
    std::fclose(f);
    auto model = std::make_shared<Acrobot>();
    model->u_min(0) = -1.5;
    model->u_max(0) = 1.5;
    ShardedBatchILQR solver(model, B, T, 0.02, devices);
    solver.init_traj(x0, u0);
    solver.iterate(iters);
    const std::vector<double> cost = solver.cost();  // the one exchange of the path
    const std::vector<double> us = solver.controls();
    {  // self-check: shards on N distinct devices must have gathered over N RCCL ranks -- anything else is a silent fallback
      std::vector<int> d(devices);
      std::sort(d.begin(), d.end());
      const int n_distinct = (int)(std::unique(d.begin(), d.end()) - d.begin());
      if (n_distinct == ns && ns > 1 && solver.rccl_ranks() != ns) {
        std::cerr << "multi_gpu: " << ns << " shards on " << ns << " devices, but the gather ran over " << solver.rccl_ranks() << " RCCL rank(s)" << std::endl;
        return 3;
      }
    }
    std::printf("multi_gpu: %d shards, gather over %s (%d ranks), mean cost %.12g\n", solver.shards(),
                solver.rccl_ranks() ? "RCCL" : "copies", solver.rccl_ranks(), [&] {
                  double s = 0;
                  for (double c : cost) s += c;
                  return s / B;
                }());
    f = std::fopen(argv[6], "wb");
    std::fwrite(cost.data(), sizeof(double), cost.size(), f);
    std::fwrite(us.data(), sizeof(double), us.size(), f);
    std::fclose(f);
  } catch (const std::exception& e) {
    std::cerr << "multi_gpu: " << e.what() << std::endl;
    return 2;
  }
  return 0;
}
