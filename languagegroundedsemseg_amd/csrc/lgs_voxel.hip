// Voxelisation on the device (the step immediately in front of the hot path).
//
//   lgs_voxelize    points[n,3] float32 -> coords[n,4] int32 = (batch, floor(A * (x, y, z, 1)))
//                   restates /root/reference/lib/voxelizer.py:136-139
//                       homo_coords = hstack(coords, 1); coords_aug = np.floor(homo_coords @ rigid_transformation.T[:, :3])
//                   (A = the 3x4 top of the 4x4 scale/rotation/translation matrix, double like numpy's) and the batch
//                   column ME.utils.sparse_collate prepends (lib/transforms.py:421).
//   lgs_label_vote  the label rule of ME.utils.sparse_quantize(coords, feats, labels, ignore_label=...)
//                   (lib/voxelizer.py:284, downstream/insseg/datasets/voxelizer.py:149): a voxel keeps the label of its
//                   first point unless another point of the voxel disagrees, then it gets ignore_label.
// Dedup itself (first occurrence wins, surviving indices ascending) is lgs_manager_insert.
//
// Integer / index work: bit-exact against the oracle.  The affine map is evaluated in double with explicitly rounded
// multiplies and adds in a fixed order ((x*a0 + y*a1) + z*a2) + a3 -- no FMA contraction -- so that floor() sees the
// same value as the numpy restatement.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/lgs_engine.h"
#include "lgs_common.h"

namespace lgs {

struct Affine { double a[12]; };

__global__ void k_voxelize(const float *__restrict__ pts, int64_t n, Affine A, int batch, int32_t *__restrict__ coords) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = (double)pts[3 * i], y = (double)pts[3 * i + 1], z = (double)pts[3 * i + 2];
  int32_t o[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const double *a = A.a + 4 * r;
    const double v = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(x, a[0]), __dmul_rn(y, a[1])), __dmul_rn(z, a[2])), a[3]);
    o[r] = (int32_t)floor(v);
  }
  reinterpret_cast<int4 *>(coords)[i] = make_int4(batch, o[0], o[1], o[2]);
}

// every point that disagrees with its voxel's representative writes the same value: no race on the result
__global__ void k_label_init(const int64_t *__restrict__ labels, const int64_t *__restrict__ unique_index, int64_t nu,
                             int64_t *__restrict__ out) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < nu) out[v] = labels[unique_index[v]];
}
__global__ void k_label_vote(const int64_t *__restrict__ labels, int64_t n, const int64_t *__restrict__ unique_index,
                             const int64_t *__restrict__ inverse, int64_t ignore_label, int64_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int64_t v = inverse[i];
  if (labels[i] != labels[unique_index[v]]) out[v] = ignore_label;
}

}  // namespace lgs

using namespace lgs;

extern "C" {

int lgs_voxelize(const float *points, int64_t n, const double *affine, int batch, int32_t *coords, void *stream) {
  LGS_REQUIRE(affine && (n == 0 || (points && coords)) && n >= 0, "lgs_voxelize: bad argument");
  LGS_REQUIRE(batch >= 0 && batch < 1024, "lgs_voxelize: batch index out of range");
  if (n == 0) return 0;
  Affine A;
  for (int i = 0; i < 12; ++i) A.a[i] = affine[i];
  LGS_KLAUNCH(k_voxelize, (unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream, points, n, A, batch, coords);
  LGS_HIP(hipGetLastError());
  return 0;
}

int lgs_label_vote(const int64_t *labels, int64_t n, const int64_t *unique_index, const int64_t *inverse, int64_t n_unique,
                   int64_t ignore_label, int64_t *labels_out, void *stream) {
  LGS_REQUIRE(n >= 0 && n_unique >= 0 && (n == 0 || (labels && unique_index && inverse && labels_out)),
              "lgs_label_vote: bad argument");
  if (n == 0 || n_unique == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  LGS_KLAUNCH(k_label_init, (unsigned)((n_unique + 255) / 256), 256, 0, s, labels, unique_index, n_unique, labels_out);
  LGS_KLAUNCH(k_label_vote, (unsigned)((n + 255) / 256), 256, 0, s, labels, n, unique_index, inverse, ignore_label, labels_out);
  LGS_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
