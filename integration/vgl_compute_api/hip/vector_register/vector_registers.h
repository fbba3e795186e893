// vector "registers" of the HIP backend: VECTOR_LENGTH per-lane slots in managed memory that a device lambda indexes with its vector_index and
// the host folds afterwards -- the contract of vgl_compute_api/{gpu,multicore}/vector_register/vector_registers.h (used by algorithms/scc/scc.hpp:110-124,
// cc, sssp).  The kernels of this backend pass vector_index = lane of the 64-wide wavefront, so VECTOR_LENGTH slots are enough.
#pragma once
#include <limits>

template <typename _T>
inline _T *vgl_hip_register_create(_T _value)
{
    _T *slots = NULL;
    MemoryAPI::allocate_array(&slots, VECTOR_LENGTH);
    for (int lane = 0; lane < VECTOR_LENGTH; lane++) slots[lane] = _value;
    return slots;
}
#define VEC_REGISTER_INT(name, value) int *reg_##name = vgl_hip_register_create<int>(value)
#define VEC_REGISTER_FLT(name, value) float *reg_##name = vgl_hip_register_create<float>(value)
#define VEC_REGISTER_DBL(name, value) double *reg_##name = vgl_hip_register_create<double>(value)

template <typename _T>
inline _T register_sum_reduce(_T *_slots)
{
    _T total = 0;
    for (int lane = 0; lane < VECTOR_LENGTH; lane++) total += _slots[lane];
    return total;
}
template <typename _T>
inline _T register_max_reduce(_T *_slots)
{
    _T best = std::numeric_limits<_T>::min();       // the other backends start here too (not lowest())
    for (int lane = 0; lane < VECTOR_LENGTH; lane++) best = _slots[lane] > best ? _slots[lane] : best;
    return best;
}
template <typename _T>
inline _T register_min_reduce(_T *_slots)
{
    _T best = std::numeric_limits<_T>::max();
    for (int lane = 0; lane < VECTOR_LENGTH; lane++) best = _slots[lane] < best ? _slots[lane] : best;
    return best;
}
template <typename _T>
inline void register_free(_T *_slots) { MemoryAPI::free_array(_slots); }
