// A user's Model WITH a device twin of its own, through the facade: the host class derives from the reference's plugin
// interface (include/model.h:6-21) as usual and additionally says (device_model_id, fill_device_desc) that the library it
// is linked against carries its twin -- examples/user_model_acrobot.hpp, compiled in with -DILQR_USER_MODEL_HEADER
// (ilqr_amd._build.build_user).  Nothing of the library's sources is edited.  The same problem is solved with the shipped
// acrobot; with the shipped acrobot's parameters the two runs are identical to the last bit.
//
//   g++ -std=c++14 -I include examples/user_model.cpp -L ilqr_amd/lib -lilqr_amd_user_example -L/opt/rocm/lib -lamdhip64
#include <cstdio>
#include <cstring>
#include <vector>

#include "ilqr_amd.hpp"

using namespace ilqr_amd;

// the host side of the user's model: the virtuals are those of the shipped Acrobot (host evaluation is not used on this
// route -- every kernel runs the device twin), the two overrides hand the twin its parameters
class MyAcrobot : public Acrobot {
 public:
  MyAcrobot(const double goal[4], double Ks, double Kd) {
    for (int i = 0; i < 4; i++) params_[i] = goal[i];
    params_[4] = Ks;
    params_[5] = Kd;
  }
  int device_model_id() const override { return ILQR_MODEL_USER; }
  void fill_device_desc(ilqr_desc& d) const override {
    d.user_params = params_;
    d.n_user_params = 6;
  }

 private:
  double params_[6];
};

int main() {
  try {
    if (!ilqr_has_user_model()) {
      std::fprintf(stderr, "error: this build of the library carries no user model\n");
      return 3;
    }
    const int T = 120;
    const double goal[4] = {3.1415, 0, 0, 0};
    VectorXd x0(4);
    x0(0) = 0.3;
    x0(1) = -0.2;
    x0(2) = 0.1;
    x0(3) = 0.0;
    VecOfVecXd u0(T, VectorXd(1));
    for (int t = 0; t < T; t++) u0[t](0) = 0;
    double cost[2];
    int its[2];
    VecOfVecXd xs[2];
    for (int which = 0; which < 2; which++) {
      Model* m = which == 0 ? static_cast<Model*>(new MyAcrobot(goal, 20.0, 20.0)) : static_cast<Model*>(new Acrobot());
      m->u_min(0) = -1.5;
      m->u_max(0) = 1.5;
      iLQR solver(m, 0.02);
      solver.verbose = false;
      solver.write_csv = false;
      solver.init_traj(x0, u0);
      for (int i = 0; i < 12 && solver.status() == ILQR_RUNNING; i++) solver.step();
      cost[which] = solver.cost();
      its[which] = solver.iterations();
      xs[which] = solver.states();
    }
    bool same = cost[0] == cost[1] && its[0] == its[1];
    for (int t = 0; t <= T && same; t++)
      for (int i = 0; i < 4; i++) same = same && xs[0][t](i) == xs[1][t](i);
    std::printf("user_twin    cost %.17g iterations %d\n", cost[0], its[0]);
    std::printf("shipped_twin cost %.17g iterations %d\n", cost[1], its[1]);
    std::printf("bit_identical %d\n", same ? 1 : 0);
    return same ? 0 : 1;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
}
