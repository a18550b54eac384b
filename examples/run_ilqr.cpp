// run_ilqr -- the two canonical problems of the reference's driver (src/run_ilqr.cpp:6-65) on the
// MI355X engine, through the C++ facade:   ./run_ilqr acrobot | integrator
#include <chrono>
#include <cstring>
#include <iostream>

#include "ilqr_amd.hpp"

using namespace ilqr_amd;

int main(int argc, char* argv[]) {
  if (argc < 2 || (std::strcmp(argv[1], "acrobot") != 0 && std::strcmp(argv[1], "integrator") != 0)) {
    std::cout << "Provide command line argument 'acrobot' or 'integrator'" << std::endl;
    return 0;
  }
  const bool quiet = argc > 2 && std::strcmp(argv[2], "--quiet") == 0;
  try {
    iLQR* ilqr;
    VecOfVecXd u0;
    VectorXd x0(4);
    const double dt = 0.02;
    if (std::strcmp(argv[1], "integrator") == 0) {
      VectorXd goal(4);
      goal(0) = 1.0; goal(1) = 0.5; goal(2) = 0.0; goal(3) = 0.0;
      ilqr = new iLQR(new DoubleIntegrator(goal), dt);
      x0(0) = -1.0; x0(1) = 0.0; x0(2) = 0.0; x0(3) = -0.2;
      VectorXd u_init(2);
      u_init.setZero();
      for (int i = 0; i < 99; i++) u0.push_back(u_init);
    } else {
      ilqr = new iLQR(new Acrobot(), dt);
      x0.setZero();
      VectorXd u_init(1);
      u_init.setZero();
      for (int i = 0; i < 499; i++) u0.push_back(u_init);
    }
    ilqr->verbose = !quiet;
    std::cout << "Run iLQR!" << std::endl;
    auto start = std::chrono::system_clock::now();
    ilqr->generate_trajectory(x0, u0);
    auto now = std::chrono::system_clock::now();
    const long elapsed = (long)std::chrono::duration_cast<std::chrono::milliseconds>(now - start).count();
    std::cout << "iLQR took: " << elapsed / 1000. << " seconds." << std::endl;
    std::printf("final cost %.12g iterations %d status %d\n", ilqr->cost(), ilqr->iterations(), ilqr->status());
    delete ilqr;
  } catch (const std::exception& e) {
    std::cerr << "run_ilqr: " << e.what() << std::endl;
    return 2;
  }
  return 0;
}
