// mac.hip — placeholder until the device MAC solver lands (next commit).
#include "common.hpp"
using namespace cslam;
struct cslam_mac { int n; };
extern "C" {
#define NI(name) return fail(CSLAM_E_INVALID, name ": not implemented yet")
int cslam_mac_create(int32_t, const int32_t*, const int32_t*, const double*, int64_t, const int32_t*, const int32_t*, const double*, int64_t, int, cslam_mac_t**) { NI("cslam_mac_create"); }
int cslam_mac_destroy(cslam_mac_t*) { return CSLAM_OK; }
int cslam_mac_laplacian(cslam_mac_t*, const double*, double, int64_t*, int32_t*, int32_t*, double*) { NI("cslam_mac_laplacian"); }
int cslam_mac_fiedler(cslam_mac_t*, const double*, double, double*, double*, int32_t*) { NI("cslam_mac_fiedler"); }
int cslam_mac_grad(cslam_mac_t*, const double*, double*) { NI("cslam_mac_grad"); }
int cslam_topk_mask(const double*, int64_t, int64_t, double*) { NI("cslam_topk_mask"); }
int cslam_mac_round_tiebreak(cslam_mac_t*, const double*, int64_t, int, double*) { NI("cslam_mac_round_tiebreak"); }
int cslam_mac_fw_subset(cslam_mac_t*, const double*, int64_t, int, double, double*, double*, double*, int32_t*, double*) { NI("cslam_mac_fw_subset"); }
int cslam_mac_stats(const cslam_mac_t*, int64_t*, double*) { NI("cslam_mac_stats"); }
}
