// k_scan_mq with the operands consumed but no matrix work (results wrong; time of the memory side only).
#pragma once
#define CSLAM_MQ_MFMA(a, b, c) ([&] { asm volatile("" ::"v"(a), "v"(b)); return (c); }())
