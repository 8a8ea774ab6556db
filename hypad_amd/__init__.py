"""hypad_amd -- MI355X-native HypAD / TadGAN train + score hot path.

Host-side mirror of the reference's Python surface (aleflabo/HypAD) over the C ABI of
``libhypad_hip.so`` (include/hypad.h).  Layout mirrors the reference's module names:

    hypad_amd.models.tadgan            <- models/tadgan.py
    hypad_amd.hyperspace.hyrnn_nets    <- hyperspace/hyrnn_nets.py (MobiusLinear, mobius_linear)
    hypad_amd.hyperspace.poincare_distance
    hypad_amd.hyperspace.gmath         <- the geoopt math functions the path uses
    hypad_amd.train                    <- train.py (the three iteration functions, train_tadgan, train)
    hypad_amd.anomaly_detection        <- anomaly_detection.py (test_tadgan)
    hypad_amd.utils.anomaly_detection_utils
    hypad_amd.engine                   <- resident-data multi-signal trainer (bench / multi-GPU path)
"""
__version__ = "0.1.0"
