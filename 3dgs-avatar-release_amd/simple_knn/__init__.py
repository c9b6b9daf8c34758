"""simple_knn -- MI355X-native implementation behind the import name the reference uses
(scene/gaussian_model.py:20: `from simple_knn._C import distCUDA2`)."""
