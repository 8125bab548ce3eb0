"""models/detection.py of the reference -> adam-dehaze_amd (HIP engine; smallest honest slice)."""
from adam_dehaze_amd.detection import (DetectionModel, IntegratedDetectionSystem, FastRCNNPredictor,  # noqa: F401
                                       create_detection_model, create_integrated_system)
