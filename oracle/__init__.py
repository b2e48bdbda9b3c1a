"""CPU oracle for the detector inference hot path of Jex-y/yolo.

TEST INFRASTRUCTURE ONLY.  Nothing under ``yolo_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` use it, and there only as the checker.

PARITY UNPINNED.  The reference is TensorFlow 2 / Keras.  TensorFlow is not
installable here, the reference ships no tests, golden vectors or fixtures,
and part of its arithmetic (Conv2D, tf.image.resize,
tf.image.combined_non_max_suppression) lives in TensorFlow itself, whose
version the reference does not pin (setup.py:28).  This oracle is therefore a
restatement written from the text of the reference files, pinned only by
hand-derived known answers and by two independent restatements agreeing with
each other (numpy here, plain C in ``oracle/c``).

Modules
-------
postproc_np   numpy fp32: decode / decode_train / decode_predict /
              filter_bboxes / combined NMS / bbox_iou / bbox_giou
              (utils.py:33-194)
net_torch     torch-CPU fp32: YOLOConv / ResidialBlock / Upsample /
              Darknet53 / YoloV3 (layers.py:17-106, backbones.py:6-60,
              detectors.py:4-77)
loss_np       numpy fp32: yolo_loss / get_yolo_loss_at_scales (losses.py:10-70)
c/            plain-C restatement of decode / filter / NMS, loaded through
              ctypes by ``oracle.cref``
"""
