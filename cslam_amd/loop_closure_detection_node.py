#!/usr/bin/env python3
"""Node shell — mirror of cslam/loop_closure_detection_node.py:13-104: declares the ROS
parameters, flattens them into the dict every hot-path class receives, builds
GlobalDescriptorLoopClosureDetection and the inter-robot detection timer.  Needs rclpy (the
classes underneath do not: tests drive them with a node double)."""
from .global_descriptor_loop_closure_detection import GlobalDescriptorLoopClosureDetection

# name → default (loop_closure_detection_node.py:20-44)
PARAMETERS = [
    ('frontend.similarity_threshold', 0.9),
    ('frontend.global_descriptor_technique', 'cosplace'),
    ('frontend.netvlad.pca_checkpoint', ''),
    ('frontend.nn_checkpoint', 'models/resnet18_64.pth'),
    ('robot_id', 0),
    ('max_nb_robots', 10),
    ('frontend.inter_robot_loop_closure_budget', 5),
    ('frontend.inter_robot_detection_period_sec', 5),
    ('frontend.nb_best_matches', 10),
    ('frontend.image_crop_size', 376),
    ('frontend.intra_loop_min_inbetween_keyframes', 10),
    ('neighbor_management.max_heartbeat_delay_sec', 5.0),
    ('neighbor_management.init_delay_sec', 5.0),
    ('neighbor_management.heartbeat_period_sec', 0.5),
    ('frontend.detection_publication_period_sec', 1.0),
    ('frontend.detection_publication_max_elems_per_msg', 10),
    ('neighbor_management.enable_neighbor_monitoring', False),
    ('frontend.enable_intra_robot_loop_closures', False),
    ('frontend.global_descriptors_topic', "global_descriptors"),
    ('frontend.inter_robot_matches_topic', "inter_robot_matches"),
    ('frontend.enable_sparsification', True),
    ('frontend.use_vertex_cover_selection', True),
    ('frontend.sensor_type', "stereo"),
    ('evaluation.enable_logs', False),
    ('evaluation.enable_sparsification_comparison', False),
]
# the keys copied into the flat dict (loop_closure_detection_node.py:45-89)
FLATTENED = [n for n, _ in PARAMETERS if n not in ('frontend.netvlad.pca_checkpoint',
                                                   'frontend.global_descriptors_topic',
                                                   'frontend.inter_robot_matches_topic')]


def flatten_parameters(node):
    params = {name: node.get_parameter(name).value for name in FLATTENED}
    params["frontend.sensor_type"] = params["frontend.sensor_type"].lower()
    return params


def default_params(**overrides):
    """The flat dict for code that runs without a ROS parameter server."""
    params = {n: v for n, v in PARAMETERS if n in FLATTENED}
    params.update(overrides)
    return params


def main():  # pragma: no cover - needs a ROS 2 installation
    import rclpy
    from rclpy.clock import Clock
    from rclpy.node import Node

    class LoopClosureDetection(Node):
        """ Global image descriptor matching for loop closure detection """

        def __init__(self):
            super().__init__('loop_closure_detection')
            self.declare_parameters(namespace='', parameters=PARAMETERS)
            self.params = flatten_parameters(self)
            self.glcd = GlobalDescriptorLoopClosureDetection(self.params, self)
            self.inter_robot_detection_timer = self.create_timer(
                self.params['frontend.inter_robot_detection_period_sec'],
                self.glcd.detect_inter, clock=Clock())

    rclpy.init(args=None)
    lcd = LoopClosureDetection()
    lcd.get_logger().info('Initialization done.')
    rclpy.spin(lcd)
    rclpy.shutdown()


if __name__ == '__main__':
    main()
