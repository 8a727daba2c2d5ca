// k_scan_mq whose ring is never refilled (results wrong; time of the compute side only).
#pragma once
#define CSLAM_MQ_REFILL false
