from .student_model import FlowStudentModel, FrameDiffStudentModel, ResidualMLP  # noqa: F401
