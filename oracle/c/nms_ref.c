/* C restatement of the reference NMS (TEST ORACLE, not product code).
 *   oracle_cpu_nms : greedy NMS with `ovr >= thresh`, restating
 *                    pyutils/mask-faster-rcnn/lib/nms/src/nms.c:35-63 (TH types replaced by plain arrays).
 *   oracle_gpu_nms : 64x64 bitmask + host OR-reduce with `IoU > thresh`, restating
 *                    nms/src/cuda/nms_kernel.cu:16-24,56-66 and nms/src/nms_cuda.c:47-58.
 * boxes are (n,4) fp32 rows [x1,y1,x2,y2]; `order` lists the rows by descending score.
 * The reference sources need TH/THC headers (absent here), so they cannot be compiled directly; this
 * file is pinned against oracle/boxes.py::nms and the reference-generated fixtures instead. */
#include <stdlib.h>
#include <string.h>
#include <math.h>

long oracle_cpu_nms(const float* boxes, const long* order, long n, float thresh, long* keep_out) {
  unsigned char* sup = (unsigned char*)calloc((size_t)n, 1);
  float* areas = (float*)malloc(sizeof(float) * (size_t)n);
  long num = 0, _i, _j;
  for (_i = 0; _i < n; ++_i) areas[_i] = (boxes[_i * 4 + 2] - boxes[_i * 4] + 1) * (boxes[_i * 4 + 3] - boxes[_i * 4 + 1] + 1);
  for (_i = 0; _i < n; ++_i) {
    long i = order[_i];
    if (sup[i]) continue;
    keep_out[num++] = i;
    float ix1 = boxes[i * 4], iy1 = boxes[i * 4 + 1], ix2 = boxes[i * 4 + 2], iy2 = boxes[i * 4 + 3], ia = areas[i];
    for (_j = _i + 1; _j < n; ++_j) {
      long j = order[_j];
      if (sup[j]) continue;
      float xx1 = fmaxf(ix1, boxes[j * 4]), yy1 = fmaxf(iy1, boxes[j * 4 + 1]);
      float xx2 = fminf(ix2, boxes[j * 4 + 2]), yy2 = fminf(iy2, boxes[j * 4 + 3]);
      float w = fmaxf(0.0f, xx2 - xx1 + 1), h = fmaxf(0.0f, yy2 - yy1 + 1);
      float inter = w * h;
      float ovr = inter / (ia + areas[j] - inter);
      if (ovr >= thresh) sup[j] = 1;
    }
  }
  free(sup); free(areas);
  return num;
}

static float dev_iou(const float* a, const float* b) {
  float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  float inter = width * height;
  float sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1), sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return inter / (sa + sb - inter);
}

/* boxes must already be sorted by descending score; keep_out indexes the sorted list */
long oracle_gpu_nms(const float* sboxes, long n, float thresh, long* keep_out) {
  long cb = (n + 63) / 64, num = 0, i, j;
  unsigned long long* remv = (unsigned long long*)calloc((size_t)cb, 8);
  for (i = 0; i < n; ++i) {
    long nb = i / 64, ib = i % 64;
    if (remv[nb] & (1ULL << ib)) continue;
    keep_out[num++] = i;
    for (j = i + 1; j < n; ++j)
      if (dev_iou(sboxes + i * 4, sboxes + j * 4) > thresh) remv[j / 64] |= 1ULL << (j % 64);
  }
  free(remv);
  return num;
}
