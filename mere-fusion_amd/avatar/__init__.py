"""Avatar preparation on the GPU: the S3FD face detector and the BiSeNet face parser as drop-ins of the reference's
module classes (face_detection/detection/sfd/net_s3fd.py, musetalk/utils/face_parsing/model.py), built on the static-graph C ABI (mf_net_*)."""
from .s3fd import s3fd                 # noqa: F401
from .bisenet import BiSeNet           # noqa: F401
