// Empty stand-in so that the reference layer sources compile in place (oracle/ref_build.sh); nothing is used.
#pragma once
