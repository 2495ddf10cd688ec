"""yolov7-tracker hot path, MI355X-native."""
__version__ = "0.1.0"
