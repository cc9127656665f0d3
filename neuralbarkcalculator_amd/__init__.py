"""MI355X-native FCN-ResNet-50 inference path for NeuralBarkCalculator."""
__version__ = "0.1.0"
