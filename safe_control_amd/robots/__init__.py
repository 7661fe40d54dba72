from .spec import RobotHandle, complete_robot_spec  # noqa: F401
