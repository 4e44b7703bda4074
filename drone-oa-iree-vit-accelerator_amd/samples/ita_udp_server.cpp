// ita_udp_server.cpp -- MI355X counterpart of the reference's UDP inference host
// (samples/inference_udp_FPGA_custom_dispatch/main.cpp:90-237), SURVEY.md section 8(f) row n1.
//
// Same wire protocol (ita_wire.h: 5 424-byte packet in, 12-byte velocity reply out, port 10001),
// same per-frame pipeline (u8 frame -> /255 -> module.main_graph -> clip/normalise/scale -> reply),
// same state handling (LSTM (h, c) carried from frame to frame) -- but for MANY senders at once:
// every sender address is a stream with its own state slot on the GPU; all packets already queued on
// the socket are served by ONE batched ita_vitlstm_forward_slots call (one stream appears at most
// once per batch, so per-stream frame order is preserved).
//
//   ita_udp_server --blob weights.itaw [--port 10001] [--device 0] [--max-streams 1024]
//                  [--max-batch 256] [--max-packets N] [--quat-stride-bug] [--quiet]
//
// Build: hipcc -O2 -std=c++17 -I include samples/ita_udp_server.cpp -L csrc -lita_mi355x -Wl,-rpath,'$ORIGIN/../csrc'
#include <arpa/inet.h>
#include <hip/hip_runtime.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>

#include "../../include/ita_mi355x.h"
#include "../../include/ita_wire.h"

#define HIPOK(e)                                                                                   \
  do {                                                                                             \
    hipError_t e_ = (e);                                                                           \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); return 2; }    \
  } while (0)
#define ITAOK(e)                                                                                   \
  do {                                                                                             \
    if ((e) != ITA_OK) { fprintf(stderr, "%s: ita status %d: %s\n", #e, ita_last_error(), ita_error_string()); return 3; } \
  } while (0)

struct Pending {
  sockaddr_in addr;
  ita_wire_frame frame;
  int slot;
};

int main(int argc, char** argv) {
  std::string blob_path;
  int port = ITA_WIRE_PORT, device = 0, max_streams = 1024, max_batch = 256, quat_bug = 0, quiet = 0;
  long max_packets = -1;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : ""; };
    if (a == "--blob") blob_path = next();
    else if (a == "--port") port = atoi(next());
    else if (a == "--device") device = atoi(next());
    else if (a == "--max-streams") max_streams = atoi(next());
    else if (a == "--max-batch") max_batch = atoi(next());
    else if (a == "--max-packets") max_packets = atol(next());
    else if (a == "--quat-stride-bug") quat_bug = 1;
    else if (a == "--quiet") quiet = 1;
    else { fprintf(stderr, "unknown argument %s\n", a.c_str()); return 1; }
  }
  if (blob_path.empty()) { fprintf(stderr, "usage: ita_udp_server --blob weights.itaw [--port N] ...\n"); return 1; }
  std::ifstream f(blob_path, std::ios::binary);
  std::vector<char> blob((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (blob.empty()) { fprintf(stderr, "cannot read %s\n", blob_path.c_str()); return 1; }

  HIPOK(hipSetDevice(device));
  ita_handle ita = nullptr;
  ITAOK(ita_create(&ita, device));
  ITAOK(ita_load_weights(ita, blob.data(), blob.size()));
  ITAOK(ita_reserve(ita, max_batch));

  // persistent per-stream LSTM state (zero = a fresh stream, like the reference's initial views)
  float *d_h, *d_c, *d_dv, *d_q, *d_vel;
  uint8_t* d_img;
  int* d_slot;
  const size_t state_bytes = sizeof(float) * 3 * (size_t)max_streams * 128;
  HIPOK(hipMalloc(&d_h, state_bytes)); HIPOK(hipMalloc(&d_c, state_bytes));
  HIPOK(hipMemset(d_h, 0, state_bytes)); HIPOK(hipMemset(d_c, 0, state_bytes));
  HIPOK(hipMalloc(&d_img, (size_t)max_batch * ITA_WIRE_IMAGE_BYTES));
  HIPOK(hipMalloc(&d_dv, sizeof(float) * max_batch)); HIPOK(hipMalloc(&d_q, sizeof(float) * 4 * max_batch));
  HIPOK(hipMalloc(&d_vel, sizeof(float) * 3 * max_batch)); HIPOK(hipMalloc(&d_slot, sizeof(int) * max_batch));
  uint8_t* h_img; float *h_dv, *h_q, *h_vel; int* h_slot;
  HIPOK(hipHostMalloc(&h_img, (size_t)max_batch * ITA_WIRE_IMAGE_BYTES));
  HIPOK(hipHostMalloc(&h_dv, sizeof(float) * max_batch)); HIPOK(hipHostMalloc(&h_q, sizeof(float) * 4 * max_batch));
  HIPOK(hipHostMalloc(&h_vel, sizeof(float) * 3 * max_batch)); HIPOK(hipHostMalloc(&h_slot, sizeof(int) * max_batch));
  hipStream_t stream;
  HIPOK(hipStreamCreate(&stream));

  const int sock = socket(AF_INET, SOCK_DGRAM, 0);
  if (sock < 0) { perror("socket"); return 1; }
  int one = 1;
  setsockopt(sock, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  int rcvbuf = 8 << 20;
  setsockopt(sock, SOL_SOCKET, SO_RCVBUF, &rcvbuf, sizeof rcvbuf);
  sockaddr_in addr{};
  addr.sin_family = AF_INET; addr.sin_addr.s_addr = INADDR_ANY; addr.sin_port = htons((uint16_t)port);
  if (bind(sock, (sockaddr*)&addr, sizeof addr) < 0) { perror("bind"); return 1; }
  if (!quiet) printf("ita_udp_server: listening on UDP %d, device %d, up to %d streams\n", port, device, max_streams);
  fflush(stdout);

  std::map<std::pair<uint32_t, uint16_t>, int> slot_of;   // sender (ip, port) -> state slot
  std::vector<std::vector<uint8_t>> packets(max_batch, std::vector<uint8_t>(ITA_WIRE_PACKET_BYTES + 16));
  std::vector<Pending> batch;
  std::vector<uint8_t> carry(ITA_WIRE_PACKET_BYTES + 16);
  sockaddr_in carry_addr{};
  bool have_carry = false;
  long served = 0, batches = 0, evicted = 0;
  std::vector<unsigned long long> last_seen(max_streams, 0);   // logical time of a slot's latest packet (LRU eviction)
  unsigned long long tick = 0;
  std::vector<int> reset_slots;                                 // slots handed to a new sender: state must restart at zero

  while (max_packets < 0 || served < max_packets) {
    batch.clear();
    std::vector<char> seen(max_streams, 0);
    // first packet: blocking; the rest: whatever is already queued (no added latency)
    while ((int)batch.size() < max_batch) {
      sockaddr_in from{};
      socklen_t flen = sizeof from;
      ssize_t n;
      uint8_t* buf = packets[batch.size()].data();
      if (have_carry) {
        memcpy(buf, carry.data(), ITA_WIRE_PACKET_BYTES); from = carry_addr; n = ITA_WIRE_PACKET_BYTES; have_carry = false;
      } else {
        n = recvfrom(sock, buf, ITA_WIRE_PACKET_BYTES + 16, batch.empty() ? 0 : MSG_DONTWAIT, (sockaddr*)&from, &flen);
        if (n < 0) {
          if (errno == EAGAIN || errno == EWOULDBLOCK) break;
          if (errno == EINTR) continue;
          perror("recvfrom");   // the reference logs and keeps serving (main.cpp:161-164)
          continue;
        }
      }
      Pending p;
      p.addr = from;
      if (ita_wire_unpack(buf, (size_t)n, quat_bug, &p.frame)) {
        if (!quiet) fprintf(stderr, "dropping short packet (%zd bytes)\n", n);
        continue;
      }
      const auto key = std::make_pair((uint32_t)from.sin_addr.s_addr, (uint16_t)from.sin_port);
      auto it = slot_of.find(key);
      if (it == slot_of.end()) {
        int slot = (int)slot_of.size();
        if (slot >= max_streams) {
          // table full: take the slot of the stream that has been silent longest (and is not in this batch);
          // its LSTM state is zeroed below, like a fresh sender's -- the reference host restarts with zero state too
          auto victim = slot_of.end();
          for (auto s2 = slot_of.begin(); s2 != slot_of.end(); ++s2)
            if (!seen[s2->second] && (victim == slot_of.end() || last_seen[s2->second] < last_seen[victim->second])) victim = s2;
          if (victim == slot_of.end()) { if (!quiet) fprintf(stderr, "every stream slot is in this batch, dropping\n"); continue; }
          slot = victim->second;
          slot_of.erase(victim);
          reset_slots.push_back(slot);
          ++evicted;
        }
        it = slot_of.emplace(key, slot).first;
      }
      p.slot = it->second;
      last_seen[p.slot] = ++tick;
      if (seen[p.slot]) {   // second frame of the same stream: it needs the state this batch produces
        memcpy(carry.data(), buf, ITA_WIRE_PACKET_BYTES); carry_addr = from; have_carry = true;
        break;
      }
      seen[p.slot] = 1;
      batch.push_back(p);
    }
    for (int slot : reset_slots)   // (3, max_streams, 128) state arrays: zero this slot's row in each layer
      for (int l = 0; l < 3; ++l) {
        HIPOK(hipMemsetAsync(d_h + ((size_t)l * max_streams + slot) * 128, 0, sizeof(float) * 128, stream));
        HIPOK(hipMemsetAsync(d_c + ((size_t)l * max_streams + slot) * 128, 0, sizeof(float) * 128, stream));
      }
    reset_slots.clear();
    if (batch.empty()) continue;
    const int B = (int)batch.size();
    for (int b = 0; b < B; ++b) {
      memcpy(h_img + (size_t)b * ITA_WIRE_IMAGE_BYTES, batch[b].frame.image, ITA_WIRE_IMAGE_BYTES);
      h_dv[b] = batch[b].frame.desired_velocity / 10.0f;   // main.cpp:179
      memcpy(h_q + 4 * b, batch[b].frame.quaternion, 16);
      h_slot[b] = batch[b].slot;
    }
    HIPOK(hipMemcpyAsync(d_img, h_img, (size_t)B * ITA_WIRE_IMAGE_BYTES, hipMemcpyHostToDevice, stream));
    HIPOK(hipMemcpyAsync(d_dv, h_dv, sizeof(float) * B, hipMemcpyHostToDevice, stream));
    HIPOK(hipMemcpyAsync(d_q, h_q, sizeof(float) * 4 * B, hipMemcpyHostToDevice, stream));
    HIPOK(hipMemcpyAsync(d_slot, h_slot, sizeof(int) * B, hipMemcpyHostToDevice, stream));
    ITAOK(ita_vitlstm_forward_slots(ita, d_img, ITA_IMAGE_U8, d_dv, d_q, d_h, d_c, d_slot, max_streams, d_vel, B, stream));
    HIPOK(hipMemcpyAsync(h_vel, d_vel, sizeof(float) * 3 * B, hipMemcpyDeviceToHost, stream));
    HIPOK(hipStreamSynchronize(stream));
    for (int b = 0; b < B; ++b) {
      float out[3];
      uint8_t reply[ITA_WIRE_REPLY_BYTES];
      ita_wire_final_velocity(h_vel + 3 * b, batch[b].frame.desired_velocity, batch[b].frame.position_x, out);
      ita_wire_pack_reply(out, reply);
      if (sendto(sock, reply, sizeof reply, 0, (sockaddr*)&batch[b].addr, sizeof(sockaddr_in)) < 0) perror("sendto");
    }
    served += B;
    ++batches;
  }
  if (!quiet) printf("ita_udp_server: served %ld packets in %ld batches from %zu streams (%ld evictions)\n", served, batches, slot_of.size(), evicted);
  close(sock);
  ita_destroy(ita);
  return 0;
}
