// lru_train_scores.h -- item GEMM + cross-entropy of the training step with stored logits (lru_train_scores.hip)
#pragma once
#include "lr_common.h"

size_t lr_train_scores_ws_floats(int R, int C);
// loss sum += into scal[0] (scal[1] = number of labelled rows, already counted); dX [R][64] is added to
// (pre-zeroed by the caller: the item splits of the d x pass use atomics);
// dE [C][64] and dbias [C] are added to with atomics (pre-zeroed gradient buffer)
int lr_launch_train_scores(const float* x, const float* E, const float* bias, const long long* labels, int R, int C, float* ws,
                           float* scal, float* dX, float* dE, float* dbias, hipStream_t st);
