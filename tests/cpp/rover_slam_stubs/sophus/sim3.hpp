#pragma once
#include "../Eigen/Core"
namespace Sophus { template <class T> class Sim3 {}; typedef Sim3<float> Sim3f; }
