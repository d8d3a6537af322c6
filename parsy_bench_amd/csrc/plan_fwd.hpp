// What host-only translation units may know of a plan (executor.hpp needs the HIP headers).
#pragma once
#include "schedule.hpp"

struct parsy_plan;
struct parsy_dist;
namespace parsy {
struct Dist;
const Schedule& plan_schedule(const parsy_plan* plan);
}
const parsy::Dist& parsy_dist_cxx(const parsy_dist* d);
