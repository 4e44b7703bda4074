/*
 * ita_wire.h -- the reference host's UDP wire format and velocity post-processing
 * (SURVEY.md section 8(f) row n1).  Plain C, no dependencies; used by samples/ita_udp_server.cpp and
 * exported from libita_mi355x.so (ita_wire_*) so the parity tests can drive it through ctypes.
 *
 * Reference: samples/inference_udp_FPGA_custom_dispatch/main.cpp
 *   :33-48    packet = 5400 u8 image | f32 desired_velocity | f32 position_x | 4 x f32 quaternion (wxyz),
 *             floats big-endian, 5424 bytes; reply = 3 x f32 = 12 bytes
 *   :320-354  unpack_frame  (quaternion loop advances by sizeof(QUAT_SIZE / 4) == 8, not 4: elements
 *             1..3 are read from the wrong offsets, two of them past the end of the packet)
 *   :356-370  reply floats are copied in HOST byte order (htonf_noswap), not swapped
 *   :179      the model's additional_data is desired_velocity / 10 (the graph divides by 10 again)
 *   :381-417  calculate_final_velocity
 */
#ifndef ITA_WIRE_H_
#define ITA_WIRE_H_

#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ITA_WIRE_IMAGE_BYTES 5400
#define ITA_WIRE_PACKET_BYTES 5424
#define ITA_WIRE_REPLY_BYTES 12
#define ITA_WIRE_PORT 10001

typedef struct ita_wire_frame {
  const uint8_t* image;       /* points into the packet: 60 x 90 u8 */
  float desired_velocity;
  float position_x;
  float quaternion[4];
} ita_wire_frame;

static inline float ita_wire_be_f32(const uint8_t* p) {
  const uint32_t u = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
  float f;
  memcpy(&f, &u, 4);
  return f;
}

/* quat_stride_bug = 0: quaternion elements at 5408 + 4*i (what the sender packs);
 * quat_stride_bug = 1: at 5408 + 8*i like the reference's loop, bytes beyond the packet read as 0. */
static inline int ita_wire_unpack(const uint8_t* packet, size_t nbytes, int quat_stride_bug, ita_wire_frame* out) {
  if (!packet || !out || nbytes < ITA_WIRE_PACKET_BYTES) return -1;
  out->image = packet;
  out->desired_velocity = ita_wire_be_f32(packet + 5400);
  out->position_x = ita_wire_be_f32(packet + 5404);
  for (int i = 0; i < 4; ++i) {
    const size_t off = 5408 + (size_t)(quat_stride_bug ? 8 : 4) * i;
    uint8_t tmp[4] = {0, 0, 0, 0};
    for (int k = 0; k < 4; ++k)
      if (off + k < nbytes) tmp[k] = packet[off + k];
    out->quaternion[i] = ita_wire_be_f32(tmp);
  }
  return 0;
}

/* main.cpp:381-417, same operation order */
static inline void ita_wire_final_velocity(const float raw[3], float desired_vel, float pos_x, float out[3]) {
  float v0 = fminf(fmaxf(raw[0], -1.0f), 1.0f), v1 = raw[1], v2 = raw[2];
  const float norm = sqrtf(v0 * v0 + v1 * v1 + v2 * v2);
  if (norm > 0.0f) { v0 /= norm; v1 /= norm; v2 /= norm; }
  v0 *= desired_vel; v1 *= desired_vel; v2 *= desired_vel;
  if (pos_x < 2.0f) v0 = fmaxf(1.0f, (pos_x / 2.0f) * desired_vel);
  out[0] = v0; out[1] = v1; out[2] = v2;
}

static inline void ita_wire_pack_reply(const float v[3], uint8_t reply[ITA_WIRE_REPLY_BYTES]) {
  memcpy(reply, v, ITA_WIRE_REPLY_BYTES);   /* host byte order, like the reference */
}

#ifdef __cplusplus
}
#endif
#endif /* ITA_WIRE_H_ */
