"""models/classifier.py of the reference -> adam-dehaze_amd (HIP engine)."""
from adam_dehaze_amd.classifier import DenseFeatureExtractor, FogIntensityClassifier, create_classifier  # noqa: F401
