"""ORACLE TEST INFRASTRUCTURE — stand-in for `cv2` (models/sam.py:10; SAM is replaced by box masks)."""
