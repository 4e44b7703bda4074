"""MI355X-native int8 ViT+LSTM engine for the ITA-dispatched hot path of
OpenHardware-Initiative/Drone-OA-IREE-ViT-Accelerator.

Sub-modules
    synth   deterministic synthetic parameters / frames (the reference ships no weights)
    params  "ITAW0001" weight+scale blob packer (include/ita_weights.h)
    host    ctypes binding of csrc/libita_mi355x.so + the host-side mirror of the reference's
            model interface (``ITAViTLSTM.forward([img, desvel, quat, (h, c)])``)
"""
__version__ = "0.1.0"
